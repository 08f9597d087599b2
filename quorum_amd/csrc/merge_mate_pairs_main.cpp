// merge_mate_pairs -- interleaves paired read files for quorum's paired-end pipeline
// (src/quorum.in:172-191: `merge_mate_pairs f1 f2 ... | error_correct_reads /dev/fd/0 | split_mate_pairs`).
//
// Mirrors src/merge_mate_pairs.cc:62-91 and src/merge_mate_pairs_cmdline.yaggo: an even number of
// files; the files at even positions form one stream, those at odd positions the other
// (skip_iterator, :9-45); records are read with jellyfish::whole_sequence_parser's rules (FASTA or
// FASTQ, multi-line sequences and qualities concatenated) and written alternately as 4-line FASTQ,
// a FASTA record getting '*' qualities (print_seq, :50-59). "Input files are not paired reads." when
// one stream ends before the other. Host-only: no GPU work on this path.
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <stdexcept>
#include <string>
#include <vector>

// A record as print_seq needs it: three spans. They point into the strings below, or -- a strict 4-line FASTQ record
// that lies whole in the stream's buffer -- at the text where it was read (valid until the stream's next call).
struct record {
  std::string header, seq, qual;
  const char *hp = nullptr, *sp = nullptr, *qp = nullptr;
  size_t hn = 0, sn = 0, qn = 0;
  void from_strings() { hp = header.data(); hn = header.size(); sp = seq.data(); sn = seq.size(); qp = qual.data(); qn = qual.size(); }
};

// one stream = several files read one after the other, in blocks of 4 MiB cut into lines with memchr (the first
// version went through std::ifstream / std::getline: 0.43 GB/s, a fifth of what the corrector behind the pipe takes)
class stream {
  std::vector<const char*> files_;
  size_t next_ = 0;
  int fd_ = -1;
  std::vector<char> buf_;
  size_t pos_ = 0, end_ = 0;
  bool file_eof_ = true;

  bool fill() {  // more bytes of the current file behind [pos_, end_); false at its end
    if (file_eof_) return false;
    if (pos_ > 0) { memmove(buf_.data(), buf_.data() + pos_, end_ - pos_); end_ -= pos_; pos_ = 0; }
    if (end_ == buf_.size()) buf_.resize(buf_.size() * 2);  // a line longer than the buffer
    for (;;) {
      const ssize_t r = read(fd_, buf_.data() + end_, buf_.size() - end_);
      if (r < 0) { if (errno == EINTR) continue; throw std::runtime_error("read error"); }
      if (r == 0) { file_eof_ = true; return false; }
      end_ += (size_t)r;
      return true;
    }
  }
  int peek() {  // the next character of the current file, EOF at its end
    if (pos_ == end_ && !fill()) return EOF;
    return (unsigned char)buf_[pos_];
  }
  // std::getline: the next line without its newline appended to `to` (the last line of a file may lack the newline);
  // false when the file has ended
  bool getline(std::string& to) {
    if (pos_ == end_ && !fill()) return false;
    for (;;) {
      const char* nl = (const char*)memchr(buf_.data() + pos_, '\n', end_ - pos_);
      if (nl) {
        to.append(buf_.data() + pos_, (size_t)(nl - (buf_.data() + pos_)));
        pos_ = (size_t)(nl - buf_.data()) + 1;
        return true;
      }
      if (!fill()) {  // the file ends inside the line
        to.append(buf_.data() + pos_, end_ - pos_);
        pos_ = end_;
        return true;
      }
    }
  }
  bool open_next() {
    while (next_ < files_.size()) {
      if (fd_ >= 0) close(fd_);
      fd_ = open(files_[next_++], O_RDONLY);
      if (fd_ < 0) throw std::runtime_error(std::string("Can't open file '") + files_[next_ - 1] + "'");
      pos_ = end_ = 0;
      file_eof_ = false;
      if (peek() != EOF) return true;
    }
    return false;
  }
  // The common case without a copy: "@header\nSEQ\n+...\nQUAL\n" followed by '@' (or the end of the file), all of it in
  // the buffer, SEQ and QUAL of one length and SEQ not empty. Anything else -- multi-line records, FASTA, a record at
  // the end of the buffer, every error -- is left to the general reader below, which decides as before.
  bool next_strict(record& r) {
    for (;;) {
      const char* b = buf_.data() + pos_;
      const char* e = buf_.data() + end_;
      const char* l1 = (const char*)memchr(b, '\n', (size_t)(e - b));
      const char* l2 = l1 ? (const char*)memchr(l1 + 1, '\n', (size_t)(e - (l1 + 1))) : nullptr;
      const char* l3 = l2 ? (const char*)memchr(l2 + 1, '\n', (size_t)(e - (l2 + 1))) : nullptr;
      const char* l4 = l3 ? (const char*)memchr(l3 + 1, '\n', (size_t)(e - (l3 + 1))) : nullptr;
      if (!l4 || (l4 + 1 == e && !file_eof_)) {  // not whole in the buffer (or its successor is not)
        if (file_eof_ || !fill()) return false;
        continue;
      }
      if (b[0] != '@' || l2[1] != '+') return false;
      const size_t sn = (size_t)(l2 - (l1 + 1)), qn = (size_t)(l4 - (l3 + 1));
      if (sn == 0 || sn != qn) return false;
      if (l4 + 1 != e && l4[1] != '@') return false;
      if (l1[1] == '+') return false;  // the "sequence" is itself a separator line: the general reader's business
      r.hp = b + 1; r.hn = (size_t)(l1 - (b + 1));
      r.sp = l1 + 1; r.sn = sn;
      r.qp = l3 + 1; r.qn = qn;
      pos_ = (size_t)(l4 + 1 - buf_.data());
      return true;
    }
  }
public:
  explicit stream(const std::vector<const char*>& f) : files_(f), buf_((size_t)4 << 20) { }
  ~stream() { if (fd_ >= 0) close(fd_); }
  bool next(record& r) {
    if (fd_ < 0 || peek() == EOF) { if (!open_next()) return false; }
    if (next_strict(r)) return true;
    const bool got = next_general(r);
    if (got) r.from_strings();
    return got;
  }
private:
  bool next_general(record& r) {
    const int c = peek();
    r.header.clear(); r.seq.clear(); r.qual.clear();
    std::string line;
    if (c == '>') {  // read_fasta: sequence lines up to the next '>'
      ++pos_;
      getline(r.header);
      while (peek() != EOF && peek() != '>') getline(r.seq);
    } else if (c == '@') {  // read_fastq
      ++pos_;
      getline(r.header);
      while (peek() != EOF && peek() != '+') getline(r.seq);
      if (peek() == EOF) throw std::runtime_error("Truncated fastq file");
      getline(line);  // the '+' line
      while (r.qual.size() < r.seq.size() && getline(r.qual)) { }
      if (r.qual.size() != r.seq.size()) throw std::runtime_error("Invalid fastq file: wrong number of quals");
      if (peek() != EOF && peek() != '@') throw std::runtime_error("Invalid fastq file: header missing");
    } else {
      throw std::runtime_error("Unsupported format");
    }
    return true;
  }
};

// stdout through one 4 MiB buffer of our own (three short fwrite calls a record cost more than the copies they made)
static std::vector<char> g_out((size_t)4 << 20);
static size_t g_out_n = 0;
static bool g_out_failed = false;
static void out_flush() {
  size_t done = 0;
  while (done < g_out_n && !g_out_failed) {
    const ssize_t w = write(1, g_out.data() + done, g_out_n - done);
    if (w < 0) { if (errno == EINTR) continue; g_out_failed = true; break; }
    done += (size_t)w;
  }
  g_out_n = 0;
}
static inline void out_put(const char* p, size_t n) {
  if (n > g_out.size() - g_out_n) {
    out_flush();
    if (n > g_out.size()) g_out.resize(n);
  }
  memcpy(g_out.data() + g_out_n, p, n);
  g_out_n += n;
}

static void print_seq(const record& s) {
  out_put("@", 1); out_put(s.hp, s.hn); out_put("\n", 1);
  out_put(s.sp, s.sn);
  out_put("\n+\n", 3);
  if (s.qn) out_put(s.qp, s.qn);
  else { std::string stars(s.sn, '*'); out_put(stars.data(), stars.size()); }
  out_put("\n", 1);
}

int main(int argc, char* argv[]) {
  std::vector<const char*> files;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "-h") || !strcmp(argv[i], "--help")) {
      printf("Usage: merge_mate_pairs file:PATH+\n\nMerge input files\n\n"
             "Take an even number of files and interleave sequences from even and odd files.\n");
      return 0;
    }
    files.push_back(argv[i]);
  }
  if (files.size() < 2) { fprintf(stderr, "Error: Requires at least 2 arguments.\n"); return 1; }
  if (files.size() % 2 != 0) { fprintf(stderr, "Error: Must give a even number files\n"); return 1; }
  std::vector<const char*> even, odd;
  for (size_t i = 0; i < files.size(); ++i) (i % 2 ? odd : even).push_back(files[i]);
  (void)fcntl(1, F_SETPIPE_SZ, 1 << 20);  // into a pipe, normally: fewer, larger transfers
  try {
    stream es(even), os(odd);
    record e, o;
    for (;;) {
      const bool he = es.next(e), ho = os.next(o);
      if (he != ho) throw std::runtime_error("Input files are not paired reads.");
      if (!he) break;
      print_seq(e);
      print_seq(o);
    }
  } catch (const std::exception& ex) {
    out_flush();
    fprintf(stderr, "terminate called after throwing an instance of 'std::runtime_error'\n  what():  %s\n", ex.what());
    return 134;  // upstream lets the exception escape main: abort
  }
  out_flush();
  return g_out_failed ? 1 : 0;
}
