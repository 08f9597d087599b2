// quorum_create_database -- drop-in command line of Quorum's database builder on
// top of libquorum_hip.
//
// Mirrors src/create_database.cc:98-129 (main) and the switch set of
// src/create_database_cmdline.yaggo:7-33: -s/--size (uint64 with SI suffix,
// required), -m/--mer, -b/--bits (required), -q/--min-qual-value,
// -Q/--min-qual-char (conflict), -t/--threads (=1), -o/--output
// (="combined_database"), -p/--reprobe (=126), reads... (at least 1).
// Not in upstream: --gpu N selects the HIP device; -t is accepted and ignored
// (the GPU supplies the parallelism).
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/quorum_hip.h"

extern "C" size_t qh_fastq_4line_prefix(const char* buf, size_t len, int final_chunk);

static const char* kUsage =
    "Usage: quorum_create_database [options] reads:path+";

static const char* kHelp =
    "Create database of k-mers for quorum error corrector\n\n"
    "Create k-mer database for quorum_error_correct\n\n"
    "Options (default value in (), *required):\n"
    " -s, --size=uint64                       *Initial hash size\n"
    " -m, --mer=uint32                        *Mer length\n"
    " -b, --bits=uint32                       *Bits for value field\n"
    " -q, --min-qual-value=uint32              Min quality as an int\n"
    " -Q, --min-qual-char=string               Min quality as a ASCII character\n"
    " -t, --threads=uint32                     Number of threads (1)\n"
    " -o, --output=path                        Output file (combined_database)\n"
    " -p, --reprobe=int32                      Maximum number of reprobes (126)\n"
    "     --gpu=int                            HIP device to build on (0)\n"
    " -U, --usage                              Usage\n"
    " -h, --help                               This message\n"
    " -V, --version                            Version";

[[noreturn]] static void error(const std::string& msg) {
  // same shape as the yaggo-generated error(): message, usage, exit(1)
  fprintf(stderr, "Error: %s\n%s\nUse --help for more information\n", msg.c_str(), kUsage);
  exit(EXIT_FAILURE);
}

static uint64_t conv_uint(const char* s, bool si_suffix, const char* what, uint64_t max) {
  if (!s || !*s || *s == '-') error(std::string("Invalid conversion of '") + (s ? s : "") + "' to " + what);
  errno = 0;
  char* end = nullptr;
  unsigned long long v = strtoull(s, &end, 0);
  if (errno || end == s) error(std::string("Invalid conversion of '") + s + "' to " + what);
  if (*end) {
    uint64_t mul = 0;
    if (si_suffix && end[1] == 0) {
      switch (*end) {
        case 'k': mul = 1000ull; break;
        case 'M': mul = 1000000ull; break;
        case 'G': mul = 1000000000ull; break;
        case 'T': mul = 1000000000000ull; break;
        case 'P': mul = 1000000000000000ull; break;
        case 'E': mul = 1000000000000000000ull; break;
      }
    }
    if (!mul || (v && mul > UINT64_MAX / v)) error(std::string("Invalid conversion of '") + s + "' to " + what);
    v *= mul;
  }
  if (v > max) error(std::string("Invalid conversion of '") + s + "' to " + what + ": out of range");
  return v;
}

struct args_t {
  uint64_t size = 0; bool size_given = false;
  uint32_t mer = 0; bool mer_given = false;
  uint32_t bits = 0; bool bits_given = false;
  uint32_t min_qual_value = 0; bool min_qual_value_given = false;
  std::string min_qual_char; bool min_qual_char_given = false;
  uint32_t threads = 1;
  const char* output = "combined_database";
  int32_t reprobe = 126;
  int gpu = 0;
  std::vector<const char*> reads;
};

static void parse(int argc, char** argv, args_t& a) {
  struct opt { char s; const char* l; };
  static const opt opts[] = {{'s', "size"}, {'m', "mer"}, {'b', "bits"}, {'q', "min-qual-value"},
                             {'Q', "min-qual-char"}, {'t', "threads"}, {'o', "output"}, {'p', "reprobe"},
                             {'G', "gpu"}};
  bool only_pos = false;
  for (int i = 1; i < argc; ++i) {
    const char* s = argv[i];
    if (only_pos || s[0] != '-' || s[1] == 0) { a.reads.push_back(s); continue; }
    if (!strcmp(s, "--")) { only_pos = true; continue; }
    if (!strcmp(s, "-h") || !strcmp(s, "--help")) { printf("%s\n\n%s\n", kUsage, kHelp); exit(0); }
    if (!strcmp(s, "-U") || !strcmp(s, "--usage")) { printf("%s\n", kUsage); exit(0); }
    if (!strcmp(s, "-V") || !strcmp(s, "--version")) { printf("%s\n", qh_version()); exit(0); }
    char key = 0;
    const char* val = nullptr;
    if (s[1] == '-') {
      const char* eq = strchr(s + 2, '=');
      const std::string name(s + 2, eq ? (size_t)(eq - s - 2) : strlen(s + 2));
      for (auto& o : opts) if (name == o.l) key = o.s;
      if (!key) error(std::string("unrecognized option '") + s + "'");
      val = eq ? eq + 1 : nullptr;
    } else {
      for (auto& o : opts) if (s[1] == o.s && o.s != 'G') key = o.s;
      if (!key) error(std::string("invalid option -- '") + s[1] + "'");
      val = s[2] ? s + 2 : nullptr;
    }
    if (!val) {
      if (i + 1 >= argc) error(std::string("option '") + s + "' requires an argument");
      val = argv[++i];
    }
    switch (key) {
      case 's': a.size = conv_uint(val, true, "uint64_t", UINT64_MAX); a.size_given = true; break;
      case 'm': a.mer = (uint32_t)conv_uint(val, false, "uint32_t", UINT32_MAX); a.mer_given = true; break;
      case 'b': a.bits = (uint32_t)conv_uint(val, false, "uint32_t", UINT32_MAX); a.bits_given = true; break;
      case 'q': a.min_qual_value = (uint32_t)conv_uint(val, false, "uint32_t", UINT32_MAX); a.min_qual_value_given = true; break;
      case 'Q': a.min_qual_char = val; a.min_qual_char_given = true; break;
      case 't': a.threads = (uint32_t)conv_uint(val, false, "uint32_t", UINT32_MAX); break;
      case 'o': a.output = val; break;
      case 'p': {
        errno = 0; char* end = nullptr; long v = strtol(val, &end, 0);
        if (errno || end == val || *end || v < INT32_MIN || v > INT32_MAX) error(std::string("Invalid conversion of '") + val + "' to int32_t");
        a.reprobe = (int32_t)v; break;
      }
      case 'G': a.gpu = (int)conv_uint(val, false, "int", 1024); break;
    }
  }
  if (!a.size_given) error("[-s, --size=uint64] required switch");
  if (!a.mer_given) error("[-m, --mer=uint32] required switch");
  if (!a.bits_given) error("[-b, --bits=uint32] required switch");
  if (a.min_qual_value_given && a.min_qual_char_given)
    error("Switches [-Q, --min-qual-char=string] and [-q, --min-qual-value=uint32] are mutually exclusive");
  if (a.reads.empty()) error("Requires at least 1 argument.");
}

[[noreturn]] static void die(const char* what, int rc) {
  fprintf(stderr, "%s: %s\n", what, qh_last_error());
  // upstream: an uncaught runtime_error("Hash is full") in a worker -> abort; the
  // driver only checks for a non-zero status (src/quorum.in:155-158)
  exit(rc == QH_EFULL ? 134 : EXIT_FAILURE);
}

// Feeds one read file. Fast path: cut the stream into chunks of whole 4-line
// groups and let the GPU index + validate them; the first chunk the GPU rejects
// switches the rest of the file to the sequential host parser (Jellyfish rules).
static void feed_file(qh_build* h, const char* path, size_t chunk_bytes) {
  FILE* f = !strcmp(path, "-") ? stdin : fopen(path, "rb");
  if (!f) { fprintf(stderr, "Can't open file '%s'\n", path); exit(EXIT_FAILURE); }
  std::vector<char> buf(chunk_bytes);
  size_t have = 0;
  bool fast = true, eof = false, any = false;
  while (!eof || have) {
    while (!eof && have < buf.size()) {
      size_t r = fread(buf.data() + have, 1, buf.size() - have, f);
      if (r == 0) { eof = true; break; }
      have += r;
    }
    if (have == 0) break;
    if (!any && buf[0] != '@' ) {
      fprintf(stderr, "Unsupported format in file '%s' (create_database needs FASTQ: it reads a quality for every base)\n", path);
      exit(EXIT_FAILURE);
    }
    any = true;
    size_t used = 0;
    if (fast) {
      used = qh_fastq_4line_prefix(buf.data(), have, eof);
      int rc = used ? qh_build_add_fastq(h, buf.data(), used) : QH_OK;
      if (rc == QH_OK && used) rc = qh_build_sync(h);
      if (rc == QH_EFORMAT || (used == 0 && (eof || have == buf.size()))) { fast = false; used = 0; }
      else if (rc != QH_OK) die("create_database", rc);
    }
    if (!fast) {
      int rc = qh_build_add_fastq_host_parsed(h, buf.data(), have, eof, &used);
      if (rc == QH_OK) rc = qh_build_sync(h);
      if (rc != QH_OK) die("create_database", rc);
      if (used == 0 && !eof && have == buf.size()) buf.resize(buf.size() * 2);  // a record longer than the chunk
    }
    memmove(buf.data(), buf.data() + used, have - used);
    have -= used;
    if (eof && used == 0 && have) {  // cannot happen: final chunks parse completely or fail
      fprintf(stderr, "Truncated fastq file '%s'\n", path);
      exit(EXIT_FAILURE);
    }
  }
  if (f != stdin) fclose(f);
}

int main(int argc, char* argv[]) {
  args_t args;
  parse(argc, argv, args);
  // src/create_database.cc:106-112
  if (!args.min_qual_value_given && !args.min_qual_char_given)
    error("Either a min-qual-value or min-qual-char must be provided.");
  if (args.min_qual_char_given && args.min_qual_char.size() != 1)
    error("The min-qual-char should be one ASCII character.");
  const char qual_thresh = args.min_qual_char_given ? args.min_qual_char[0] : (char)args.min_qual_value;
  if (args.bits < 1 || args.bits > 63)
    error("The number of bits should be between 1 and 63");
  // :113-115 -- open the output before doing any work
  FILE* out = fopen(args.output, "wb");
  if (!out) error(std::string("Failed to open output file '") + args.output + "'.");
  fclose(out);

  qh_build* h = nullptr;
  int rc = qh_build_create(&h, (int)args.mer, (int)args.bits, args.size, (int)qual_thresh, args.reprobe, args.gpu, 0, 0);
  if (rc != QH_OK) error(qh_last_error());

  const char* cb = getenv("QUORUM_HIP_CHUNK_MB");
  size_t chunk = (cb && atol(cb) > 0 ? (size_t)atol(cb) : 256) << 20;
  for (const char* path : args.reads) feed_file(h, path, chunk);

  qh_build_stats st;
  if ((rc = qh_build_finish(h, &st)) != QH_OK) die("create_database", rc);
  if ((rc = qh_build_write(h, args.output, argc, argv)) != QH_OK) die("create_database", rc);
  if (getenv("QUORUM_HIP_VERBOSE"))
    fprintf(stderr, "reads %llu  k-mers %llu  high-quality %llu  distinct %llu  slots %llu  doublings %llu\n",
            (unsigned long long)st.records, (unsigned long long)st.instances, (unsigned long long)st.hq_instances,
            (unsigned long long)st.distinct, (unsigned long long)st.slots, (unsigned long long)st.grows);
  qh_build_destroy(h);
  return 0;
}
