// merge_mate_pairs -- interleaves paired read files for quorum's paired-end pipeline
// (src/quorum.in:172-191: `merge_mate_pairs f1 f2 ... | error_correct_reads /dev/fd/0 | split_mate_pairs`).
//
// Mirrors src/merge_mate_pairs.cc:62-91 and src/merge_mate_pairs_cmdline.yaggo: an even number of
// files; the files at even positions form one stream, those at odd positions the other
// (skip_iterator, :9-45); records are read with jellyfish::whole_sequence_parser's rules (FASTA or
// FASTQ, multi-line sequences and qualities concatenated) and written alternately as 4-line FASTQ,
// a FASTA record getting '*' qualities (print_seq, :50-59). "Input files are not paired reads." when
// one stream ends before the other. Host-only: no GPU work on this path.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

struct record { std::string header, seq, qual; };

// one stream = several files read one after the other
class stream {
  std::vector<const char*> files_;
  size_t next_ = 0;
  std::ifstream is_;
  bool open_next() {
    while (next_ < files_.size()) {
      is_.close(); is_.clear();
      is_.open(files_[next_++]);
      if (!is_.good()) throw std::runtime_error(std::string("Can't open file '") + files_[next_ - 1] + "'");
      if (is_.peek() != EOF) return true;
    }
    return false;
  }
public:
  explicit stream(const std::vector<const char*>& f) : files_(f) { }
  bool next(record& r) {
    if (!is_.is_open() || is_.peek() == EOF) { if (!open_next()) return false; }
    const int c = is_.peek();
    r.header.clear(); r.seq.clear(); r.qual.clear();
    std::string line;
    if (c == '>') {  // read_fasta: sequence lines up to the next '>'
      is_.get();
      std::getline(is_, r.header);
      while (is_.peek() != EOF && is_.peek() != '>') { std::getline(is_, line); r.seq += line; }
    } else if (c == '@') {  // read_fastq
      is_.get();
      std::getline(is_, r.header);
      while (is_.peek() != EOF && is_.peek() != '+') { std::getline(is_, line); r.seq += line; }
      if (!is_.good()) throw std::runtime_error("Truncated fastq file");
      std::getline(is_, line);  // the '+' line
      while (r.qual.size() < r.seq.size() && is_.good()) { std::getline(is_, line); r.qual += line; }
      if (r.qual.size() != r.seq.size()) throw std::runtime_error("Invalid fastq file: wrong number of quals");
      if (is_.peek() != EOF && is_.peek() != '@') throw std::runtime_error("Invalid fastq file: header missing");
    } else {
      throw std::runtime_error("Unsupported format");
    }
    return true;
  }
};

static void print_seq(const record& s) {
  fputc('@', stdout); fwrite(s.header.data(), 1, s.header.size(), stdout); fputc('\n', stdout);
  fwrite(s.seq.data(), 1, s.seq.size(), stdout);
  fputs("\n+\n", stdout);
  if (!s.qual.empty()) fwrite(s.qual.data(), 1, s.qual.size(), stdout);
  else for (size_t i = 0; i < s.seq.size(); ++i) fputc('*', stdout);
  fputc('\n', stdout);
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
    fflush(stdout);
    fprintf(stderr, "terminate called after throwing an instance of 'std::runtime_error'\n  what():  %s\n", ex.what());
    return 134;  // upstream lets the exception escape main: abort
  }
  return ferror(stdout) ? 1 : 0;
}
