// split_mate_pairs -- the last stage of quorum's paired-end pipeline (src/quorum.in:213-221).
//
// Mirrors src/split_mate_pairs.cc:15-33 and src/split_mate_pairs_cmdline.yaggo: reads the corrector's
// FASTA stream from stdin two lines at a time (header, sequence) and writes the records alternately
// to <prefix>_1.fa and <prefix>_2.fa. The corrector is run with -d (--no-discard, quorum.in:161) so
// that every read, discarded or not, leaves exactly one two-line record and the mates stay in step.
// Host-only: no GPU work on this path.
#include <fstream>
#include <iostream>
#include <string>

int main(int argc, char* argv[]) {
  if (argc == 2 && (std::string(argv[1]) == "-h" || std::string(argv[1]) == "--help")) {
    std::cout << "Usage: split_mate_pairs prefix:PATH\n\nSplit fasta file in two\n\n"
                 "Read fasta file from stdin and write sequence alternatively to two output files\n";
    return 0;
  }
  if (argc != 2) { std::cerr << "Error: Requires exactly 1 argument.\n"; return 1; }
  const std::string file1 = std::string(argv[1]) + "_1.fa", file2 = std::string(argv[1]) + "_2.fa";
  std::ofstream out1(file1), out2(file2);
  if (!out1.good()) { std::cerr << "Error: Failed to open output file '" << file1 << "'\n"; return 1; }
  if (!out2.good()) { std::cerr << "Error: Failed to open output file '" << file2 << "'\n"; return 1; }
  std::string line;
  bool first = true;
  for (; std::getline(std::cin, line); first = !first) {
    std::ostream& os = first ? out1 : out2;
    os << line << '\n';
    std::getline(std::cin, line);
    os << line << '\n';
    if (!os.good()) { std::cerr << "Error: Error while writing to file '" << (first ? file1 : file2) << "'\n"; return 1; }
  }
  return 0;
}
