// split_mate_pairs -- the last stage of quorum's paired-end pipeline (src/quorum.in:213-221).
//
// Mirrors src/split_mate_pairs.cc:15-33 and src/split_mate_pairs_cmdline.yaggo: reads the corrector's
// FASTA stream from stdin two lines at a time (header, sequence) and writes the records alternately
// to <prefix>_1.fa and <prefix>_2.fa. The corrector is run with -d (--no-discard, quorum.in:161) so
// that every read, discarded or not, leaves exactly one two-line record and the mates stay in step.
// Host-only: no GPU work on this path. The stream is taken in blocks of 8 MiB and cut with memchr (the
// first version went through std::getline on std::cin: 48 MB/s, fifty times slower than the corrector
// that feeds it).
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <string>
#include <vector>

static bool write_all(FILE* f, const char* p, size_t n) { return n == 0 || fwrite(p, 1, n, f) == n; }

int main(int argc, char* argv[]) {
  if (argc == 2 && (std::string(argv[1]) == "-h" || std::string(argv[1]) == "--help")) {
    fputs("Usage: split_mate_pairs prefix:PATH\n\nSplit fasta file in two\n\n"
          "Read fasta file from stdin and write sequence alternatively to two output files\n", stdout);
    return 0;
  }
  if (argc != 2) { fputs("Error: Requires exactly 1 argument.\n", stderr); return 1; }
  const std::string name[2] = {std::string(argv[1]) + "_1.fa", std::string(argv[1]) + "_2.fa"};
  FILE* out[2];
  std::vector<char> obuf[2];
  for (int i = 0; i < 2; ++i) {
    out[i] = fopen(name[i].c_str(), "w");
    if (!out[i]) { fprintf(stderr, "Error: Failed to open output file '%s'\n", name[i].c_str()); return 1; }
    obuf[i].resize((size_t)4 << 20);
    setvbuf(out[i], obuf[i].data(), _IOFBF, obuf[i].size());
  }
  (void)fcntl(0, F_SETPIPE_SZ, 1 << 20);  // from a pipe, normally: fewer, larger transfers
  std::vector<char> buf((size_t)8 << 20);
  size_t have = 0;       // bytes in buf
  bool eof = false;
  int which = 0;         // the file the next record goes to
  auto fail = [&](int i) { fprintf(stderr, "Error: Error while writing to file '%s'\n", name[i].c_str()); return 1; };
  while (!eof || have) {
    if (!eof) {
      if (have == buf.size()) buf.resize(buf.size() * 2);  // a record longer than the buffer
      const ssize_t r = read(0, buf.data() + have, buf.size() - have);
      if (r < 0) { if (errno == EINTR) continue; perror("read"); return 1; }
      if (r == 0) eof = true;
      have += (size_t)r;
    }
    // whole records (two lines) go out; consecutive records are written per file in one piece each
    size_t pos = 0;
    for (;;) {
      const char* a = (const char*)memchr(buf.data() + pos, '\n', have - pos);
      const char* b = a ? (const char*)memchr(a + 1, '\n', have - (size_t)(a + 1 - buf.data())) : nullptr;
      if (!b) break;
      const size_t end = (size_t)(b - buf.data()) + 1;
      if (!write_all(out[which], buf.data() + pos, end - pos)) return fail(which);
      which ^= 1;
      pos = end;
    }
    if (eof && pos < have) {
      // the stream's last record lacks a newline, or its sequence line: both lines are written, as upstream's
      // two getline + '\n' do
      const char* a = (const char*)memchr(buf.data() + pos, '\n', have - pos);
      if (!write_all(out[which], buf.data() + pos, have - pos)) return fail(which);
      if (!write_all(out[which], "\n", 1)) return fail(which);
      if (!a && !write_all(out[which], "\n", 1)) return fail(which);
      pos = have;
    }
    memmove(buf.data(), buf.data() + pos, have - pos);
    have -= pos;
  }
  for (int i = 0; i < 2; ++i)
    if (fclose(out[i]) != 0) return fail(i);
  return 0;
}
