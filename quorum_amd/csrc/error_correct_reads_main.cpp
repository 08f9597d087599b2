// quorum_error_correct_reads -- drop-in command line of Quorum's read corrector on top of
// libquorum_hip.
//
// Mirrors main of src/error_correct_reads.cc:676-742 and the switch set of
// src/error_correct_reads_cmdline.yaggo:6-71: -t/--thread (=1), -m/--min-count (=1), -s/--skip (=1),
// -g/--good (=2), -a/--anchor-count (=3), -w/--window (=10), -e/--error (=3), -o/--output prefix,
// --contaminant path, --trim-contaminant, --homo-trim int32, --gzip, -M/--no-mmap,
// --apriori-error-rate (=0.01), --poisson-threshold (=1e-6), -p/--cutoff, -q/--qual-cutoff-value,
// -Q/--qual-cutoff-char (conflict), -d/--no-discard, -v/--verbose; positional: db, sequence+.
// Output: <prefix>.fa and <prefix>.log, or stdout / stderr when -o is not given (:160-161,:723).
// Differences from upstream: -t is accepted and ignored (the GPU supplies the parallelism; reads keep
// their input order, as upstream with -t 1); --contaminant takes the FASTA file itself (upstream takes a
// `jellyfish count -C` dump of it, Makefile.am:54-55); --gzip is refused (no zlib stream here);
// --gpu N selects the HIP device; --devices a,b,... runs one replica of the corrector per listed device (the database
// loaded on each): reads are independent, so every chunk is cut at record boundaries into as many parts, the parts are
// corrected side by side and their text is written in input order -- the same bytes as with one device.
#include <errno.h>
#include <fcntl.h>
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../../include/quorum_hip.h"

static const char* kUsage = "Usage: error_correct_reads [options] db:path sequence:path+";

static const char* kHelp =
    "Error correct reads from a fastq file based on the k-mer frequencies.\n\n"
    "Options (default value in (), *required):\n"
    " -t, --thread=uint32                      Number of threads (1)\n"
    " -m, --min-count=uint32                   Minimum count for a k-mer to be considered \"good\" (1)\n"
    " -s, --skip=uint32                        Number of bases to skip for start k-mer (1)\n"
    " -g, --good=uint32                        Number of good k-mer in a row for anchor (2)\n"
    " -a, --anchor-count=uint32                Minimum count for an anchor k-mer (3)\n"
    " -w, --window=uint32                      Size of window (10)\n"
    " -e, --error=uint32                       Maximum number of error in a window (3)\n"
    " -o, --output=prefix                      Output file prefix (error_corrected)\n"
    "     --contaminant=path                   FASTA file of contaminant sequences\n"
    "     --trim-contaminant                   Trim reads containing contaminated k-mers instead of discarding (false)\n"
    "     --homo-trim=int32                    Trim homo-polymer run at the 3' end\n"
    "     --gzip                               Gzip output file (not supported)\n"
    " -M, --no-mmap                            Do not memory map the input mer database (false)\n"
    "     --apriori-error-rate=double          Probability of a base being an error (0.01)\n"
    "     --poisson-threshold=double           Error probability threshold in Poisson test (1e-6)\n"
    " -p, --cutoff=uint32                      Poisson cutoff when there are multiple choices\n"
    " -q, --qual-cutoff-value=uint32           Any base above with quality equal or greater is untouched when there are multiple choices\n"
    " -Q, --qual-cutoff-char=string            Any base above with quality equal or greater is untouched when there are multiple choices\n"
    " -d, --no-discard                         Do not discard reads, output a single N (false)\n"
    " -v, --verbose                            Be verbose (false)\n"
    "     --gpu=int                            HIP device (0)\n"
    "     --devices=list                       HIP devices a,b,...: one replica of the corrector on each\n"
    " -U, --usage                              Usage\n"
    " -h, --help                               This message\n"
    " -V, --version                            Version";

[[noreturn]] static void error(const std::string& msg) {
  fprintf(stderr, "Error: %s\n%s\nUse --help for more information\n", msg.c_str(), kUsage);
  exit(EXIT_FAILURE);
}

[[noreturn]] static void die(const std::string& msg) {  // err::die
  fprintf(stderr, "%s\n", msg.c_str());
  exit(EXIT_FAILURE);
}

static uint32_t conv_u32(const char* s) {
  if (!s || !*s || *s == '-') error(std::string("Invalid conversion of '") + (s ? s : "") + "' to uint32_t");
  errno = 0;
  char* end = nullptr;
  unsigned long long v = strtoull(s, &end, 0);
  if (errno || end == s || *end || v > UINT32_MAX) error(std::string("Invalid conversion of '") + s + "' to uint32_t");
  return (uint32_t)v;
}
static int32_t conv_i32(const char* s) {
  errno = 0;
  char* end = nullptr;
  long v = strtol(s ? s : "", &end, 0);
  if (!s || errno || end == s || *end || v < INT32_MIN || v > INT32_MAX) error(std::string("Invalid conversion of '") + (s ? s : "") + "' to int32_t");
  return (int32_t)v;
}
static double conv_double(const char* s) {
  errno = 0;
  char* end = nullptr;
  double v = strtod(s ? s : "", &end);
  if (!s || errno || end == s || *end) error(std::string("Invalid conversion of '") + (s ? s : "") + "' to double_t");
  return v;
}

struct args_t {
  uint32_t thread = 1, min_count = 1, skip = 1, good = 2, anchor_count = 3, window = 10, error = 3;
  std::string output = "error_corrected"; bool output_given = false;
  const char* contaminant = nullptr;
  bool trim_contaminant = false, gzip = false, no_mmap = false, no_discard = false, verbose = false;
  int32_t homo_trim = 0; bool homo_trim_given = false;
  double apriori_error_rate = 0.01, poisson_threshold = 1e-6;
  uint32_t cutoff = 0; bool cutoff_given = false;
  uint32_t qual_cutoff_value = 0; bool qual_cutoff_value_given = false;
  std::string qual_cutoff_char; bool qual_cutoff_char_given = false;
  int gpu = 0;
  std::vector<int> devices;
  const char* db = nullptr;
  std::vector<const char*> sequence;
};

static void parse(int argc, char** argv, args_t& a) {
  struct opt { char s; const char* l; bool has_arg; };
  // keys without a short form use control characters
  static const opt opts[] = {
      {'t', "thread", true}, {'m', "min-count", true}, {'s', "skip", true}, {'g', "good", true},
      {'a', "anchor-count", true}, {'w', "window", true}, {'e', "error", true}, {'o', "output", true},
      {1, "contaminant", true}, {2, "trim-contaminant", false}, {3, "homo-trim", true}, {4, "gzip", false},
      {'M', "no-mmap", false}, {5, "apriori-error-rate", true}, {6, "poisson-threshold", true},
      {'p', "cutoff", true}, {'q', "qual-cutoff-value", true}, {'Q', "qual-cutoff-char", true},
      {'d', "no-discard", false}, {'v', "verbose", false}, {7, "gpu", true}, {8, "devices", true}};
  std::vector<const char*> pos;
  bool only_pos = false;
  for (int i = 1; i < argc; ++i) {
    const char* s = argv[i];
    if (only_pos || s[0] != '-' || s[1] == 0) { pos.push_back(s); continue; }
    if (!strcmp(s, "--")) { only_pos = true; continue; }
    if (!strcmp(s, "-h") || !strcmp(s, "--help")) { printf("%s\n\n%s\n", kUsage, kHelp); exit(0); }
    if (!strcmp(s, "-U") || !strcmp(s, "--usage")) { printf("%s\n", kUsage); exit(0); }
    if (!strcmp(s, "-V") || !strcmp(s, "--version")) { printf("%s\n", qh_version()); exit(0); }
    const opt* o = nullptr;
    const char* val = nullptr;
    if (s[1] == '-') {
      const char* eq = strchr(s + 2, '=');
      const std::string name(s + 2, eq ? (size_t)(eq - s - 2) : strlen(s + 2));
      for (auto& x : opts) if (name == x.l) o = &x;
      if (!o) error(std::string("unrecognized option '") + s + "'");
      val = eq ? eq + 1 : nullptr;
      if (!o->has_arg && val) error(std::string("option '--") + o->l + "' doesn't allow an argument");
    } else {
      for (auto& x : opts) if (s[1] == x.s && x.s > 8) o = &x;
      if (!o) error(std::string("invalid option -- '") + s[1] + "'");
      if (o->has_arg) val = s[2] ? s + 2 : nullptr;
      else if (s[2]) error(std::string("invalid option -- '") + s[2] + "'");
    }
    if (o->has_arg && !val) {
      if (i + 1 >= argc) error(std::string("option '") + s + "' requires an argument");
      val = argv[++i];
    }
    switch (o->s) {
      case 't': a.thread = conv_u32(val); break;
      case 'm': a.min_count = conv_u32(val); break;
      case 's': a.skip = conv_u32(val); break;
      case 'g': a.good = conv_u32(val); break;
      case 'a': a.anchor_count = conv_u32(val); break;
      case 'w': a.window = conv_u32(val); break;
      case 'e': a.error = conv_u32(val); break;
      case 'o': a.output = val; a.output_given = true; break;
      case 1: a.contaminant = val; break;
      case 2: a.trim_contaminant = true; break;
      case 3: a.homo_trim = conv_i32(val); a.homo_trim_given = true; break;
      case 4: a.gzip = true; break;
      case 'M': a.no_mmap = true; break;
      case 5: a.apriori_error_rate = conv_double(val); break;
      case 6: a.poisson_threshold = conv_double(val); break;
      case 'p': a.cutoff = conv_u32(val); a.cutoff_given = true; break;
      case 'q': a.qual_cutoff_value = conv_u32(val); a.qual_cutoff_value_given = true; break;
      case 'Q': a.qual_cutoff_char = val; a.qual_cutoff_char_given = true; break;
      case 'd': a.no_discard = true; break;
      case 'v': a.verbose = true; break;
      case 7: a.gpu = (int)conv_u32(val); break;
      case 8: {
        a.devices.clear();
        std::string item;
        for (const char* c = val;; ++c) {
          if (*c == ',' || *c == 0) { a.devices.push_back((int)conv_u32(item.c_str())); item.clear(); if (!*c) break; }
          else item += *c;
        }
        break;
      }
    }
  }
  if (a.qual_cutoff_value_given && a.qual_cutoff_char_given)
    error("Switches [-Q, --qual-cutoff-char=string] and [-q, --qual-cutoff-value=uint32] are mutually exclusive");
  if (pos.size() < 2) error("Requires at least 2 arguments.");
  a.db = pos[0];
  a.sequence.assign(pos.begin() + 1, pos.end());
}

static void write_all(int fd, const char* p, size_t n, const std::string& name) {
  while (n) {
    ssize_t w = write(fd, p, n);
    if (w < 0) { if (errno == EINTR) continue; die("Failed to write to file '" + name + "': " + strerror(errno)); }
    p += w; n -= (size_t)w;
  }
}

// Fills dst[0, n) from the file: a regular file is read by a few threads at once (pread of disjoint slices: the
// page-cache copy is what bounds the tool), anything else sequentially.
static size_t read_some(int fd, bool regular, off_t* pos, char* dst, size_t n) {
  if (!regular) {
    size_t got = 0;
    while (got < n) {
      ssize_t r = read(fd, dst + got, n - got);
      if (r < 0) { if (errno == EINTR) continue; die(std::string("read error: ") + strerror(errno)); }
      if (r == 0) break;
      got += (size_t)r;
    }
    return got;
  }
  struct stat st;
  if (fstat(fd, &st) != 0) die(std::string("fstat: ") + strerror(errno));
  const size_t left = st.st_size > *pos ? (size_t)(st.st_size - *pos) : 0;
  if (n > left) n = left;
  const size_t min_slice = (size_t)8 << 20;
  unsigned nt = std::thread::hardware_concurrency();
  if (nt > 8) nt = 8;
  if (nt < 1 || n < 2 * min_slice) nt = 1;
  const size_t slice = (n + nt - 1) / nt;
  bool failed = false;
  auto work = [&](size_t from, size_t to) {
    while (from < to) {
      ssize_t r = pread(fd, dst + from, to - from, *pos + (off_t)from);
      if (r < 0) { if (errno == EINTR) continue; failed = true; return; }
      if (r == 0) { failed = true; return; }  // file shrank under us
      from += (size_t)r;
    }
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, (size_t)t * slice, ((size_t)t + 1) * slice < n ? ((size_t)t + 1) * slice : n);
  work(0, slice < n ? slice : n);
  for (auto& t : th) t.join();
  if (failed) die("read error");
  *pos += (off_t)n;
  return n;
}

int main(int argc, char* argv[]) {
  args_t args;
  parse(argc, argv, args);
  // src/error_correct_reads.cc:680-685
  if (args.qual_cutoff_char_given && args.qual_cutoff_char.size() != 1)
    error("The qual-cutoff-char must be one ASCII character.");
  if (args.qual_cutoff_value_given && args.qual_cutoff_value > (uint32_t)CHAR_MAX)
    error("The qual-cutoff-value must be in the range 0-127.");
  const char qual_cutoff = args.qual_cutoff_char_given ? args.qual_cutoff_char[0]
                           : (args.qual_cutoff_value_given ? (char)args.qual_cutoff_value : (char)CHAR_MAX);
  if (args.gzip) error("--gzip is not supported by this build: pipe the .fa output through gzip");

  const auto t_start = std::chrono::steady_clock::now();
  auto secs = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
  auto vlog = [&](const std::string& m) { if (args.verbose) fprintf(stderr, "[%.3f] %s\n", secs(), m.c_str()); };

  if (args.devices.empty()) args.devices.push_back(args.gpu);
  const int nrep = (int)args.devices.size();
  struct replica { qh_query* db = nullptr; qh_query* contaminant = nullptr; qh_corrector* ec = nullptr; };
  std::vector<replica> rep((size_t)nrep);
  vlog("Loading mer database");  // :688-690
  for (int r = 0; r < nrep; ++r)
    if (qh_query_open(&rep[r].db, args.db, args.devices[r]) != QH_OK) die(qh_last_error());
  qh_query* db = rep[0].db;
  int k = 0, bits = 0;
  uint64_t size = 0;
  qh_query_info(db, &k, &bits, &size);

  if (args.contaminant) {  // :692-708
    vlog("Loading contaminant sequences");
    for (int r = 0; r < nrep; ++r)
      if (qh_contaminant_open_fasta(&rep[r].contaminant, args.contaminant, k, args.devices[r]) != QH_OK) die(qh_last_error());
  }

  unsigned cutoff = args.cutoff;  // :712-718
  if (!args.cutoff_given) {
    vlog("Computing Poisson cutoff");
    if (qh_query_poisson_cutoff(db, args.apriori_error_rate / 3, args.poisson_threshold / args.apriori_error_rate, &cutoff) != QH_OK)
      die(qh_last_error());
  }
  vlog("Using cutoff of " + std::to_string(cutoff));
  if (cutoff == 0 && !args.cutoff_given) die("Cutoff computation failed. Pass it explicitly with -p switch.");

  qh_correct_params p;  // :720-736
  memset(&p, 0, sizeof p);
  p.skip = (int32_t)args.skip; p.good = (int32_t)args.good; p.anchor = (int32_t)args.anchor_count;
  p.min_count = (int32_t)args.min_count; p.cutoff = (int32_t)cutoff; p.window = (int32_t)args.window;
  p.error = (int32_t)args.error; p.qual_cutoff = (int32_t)qual_cutoff; p.trim_contaminant = args.trim_contaminant;
  p.do_homo_trim = args.homo_trim_given; p.homo_trim = args.homo_trim; p.no_discard = args.no_discard;
  p.collision_prob = args.apriori_error_rate / 3; p.poisson_threshold = args.poisson_threshold;
  // The corrector's private, sparser copy of the table (QH_EC_TABLE_LOAD) makes its walks 17 % faster but has to be
  // allocated, cleared and filled first: about the time of correcting 3 bases per byte of the copy (measured: 0.13 s
  // for a 2^30-slot copy, against 0.02 s saved on 2 GB of reads). One run over the reads a database was built from
  // never gets there, so the tool asks for the copy only when its input is at least 64 bytes per slot.
  if (!getenv("QH_EC_TABLE_LOAD")) {
    uint64_t in_bytes = 0;
    bool known = true;
    for (const char* path : args.sequence) {
      struct stat ist;
      if (strcmp(path, "-") && stat(path, &ist) == 0 && S_ISREG(ist.st_mode)) in_bytes += (uint64_t)ist.st_size;
      else known = false;
    }
    if (!known || in_bytes / 64 < size) setenv("QH_EC_TABLE_LOAD", "0", 1);
  }
  for (int r = 0; r < nrep; ++r)
    if (qh_corrector_create(&rep[r].ec, rep[r].db, rep[r].contaminant, &p) != QH_OK) die(qh_last_error());
  qh_corrector* ec = rep[0].ec;

  // open_file (:134-155): <prefix>.log / <prefix>.fa, or stderr / stdout without a prefix
  const std::string prefix = args.output_given ? args.output : "";
  const std::string log_name = prefix.empty() ? "/dev/fd/2" : prefix + ".log";
  const std::string fa_name = prefix.empty() ? "/dev/fd/1" : prefix + ".fa";
  const int log_fd = prefix.empty() ? 2 : open(log_name.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
  if (log_fd < 0) die("Failed to open file '" + log_name + "': " + strerror(errno));
  const int fa_fd = prefix.empty() ? 1 : open(fa_name.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
  if (fa_fd < 0) die("Failed to open file '" + fa_name + "': " + strerror(errno));

  vlog("Correcting reads");
  (void)fcntl(fa_fd, F_SETPIPE_SZ, 1 << 20);  // when the corrected reads go down a pipe (split_mate_pairs)
  const char* cb = getenv("QUORUM_HIP_CHUNK_MB");
  const size_t chunk_bytes = (cb && atol(cb) > 0 ? (size_t)atol(cb) : 256) << 20;
  char* buf = nullptr;
  size_t cap = chunk_bytes;
  bool pinned = qh_host_alloc((void**)&buf, cap) == QH_OK;
  if (!pinned) buf = (char*)malloc(cap);
  if (!buf) die("out of memory for the chunk buffer");
  qh_correct_stats total;
  memset(&total, 0, sizeof total);
  double t_read = 0, t_correct = 0, t_write = 0;  // where the wall clock of the loop goes (-v); t_write: waiting for the writer
  // chunk i is written out by a second thread while chunk i+1 is read and corrected: what qh_corrector_fetch hands
  // out stays valid through the next correct + fetch
  std::thread writer;
  auto join_writer = [&]() { if (writer.joinable()) writer.join(); };
  // (appending through a shared mapping filled by eight threads, which pays for the database file, was measured here
  // and gains nothing over write(): the pieces are 150 MB and the loop already hides most of the write)
  // one chunk through the corrector; its text goes to the writer thread. Returns the bytes that formed whole records.
  struct piece { const char* fa; size_t nfa; const char* lg; size_t nlg; };
  auto add_stats = [&](const qh_correct_stats& st) {
    total.reads += st.reads; total.corrected += st.corrected; total.skipped += st.skipped;
    total.substitutions += st.substitutions; total.truncations += st.truncations;
    total.bases_in += st.bases_in; total.bases_out += st.bases_out;
  };
  auto process = [&](const char* chunk, size_t have, bool eof) -> size_t {
    size_t used = 0;
    const double t1 = secs();
    std::vector<piece> out;
    // Replicas: the chunk's whole 4-line records, cut into one part per device at record boundaries; every part is
    // corrected by its replica and the texts follow one another in input order. Text the cut cannot be trusted on
    // (not strict 4-line records) goes to the first replica whole, like with one device.
    bool split_done = false;
    if (nrep > 1) {
      const size_t whole = qh_fastq_4line_prefix(chunk, have, eof);
      std::vector<size_t> cut((size_t)nrep + 1, 0);
      if (whole && qh_fastq_split_records(chunk, whole, nrep, cut.data()) == QH_OK) {
        std::vector<piece> parts((size_t)nrep, piece{nullptr, 0, nullptr, 0});
        std::vector<qh_correct_stats> sts((size_t)nrep);
        std::vector<std::string> errs((size_t)nrep);
        std::vector<char> ok((size_t)nrep, 1);
        auto one = [&](int r) {
          const size_t len = cut[r + 1] - cut[r];
          memset(&sts[r], 0, sizeof sts[r]);
          if (!len) return;
          size_t u = 0;
          if (qh_corrector_correct_fastq(rep[r].ec, chunk + cut[r], len, 1, &u) != QH_OK || u != len ||
              qh_corrector_fetch(rep[r].ec, &parts[r].fa, &parts[r].nfa, &parts[r].lg, &parts[r].nlg, &sts[r]) != QH_OK) {
            ok[r] = 0;
            errs[r] = qh_last_error();
          }
        };
        std::vector<std::thread> th;
        for (int r = 1; r < nrep; ++r) th.emplace_back(one, r);
        one(0);
        for (auto& t : th) t.join();
        bool all = true;
        for (char c : ok) all = all && c;
        if (all) {
          for (int r = 0; r < nrep; ++r) { out.push_back(parts[r]); add_stats(sts[r]); }
          used = whole;
          split_done = true;
        }
      }
    }
    if (!split_done) {
      if (qh_corrector_correct_fastq(ec, chunk, have, eof, &used) != QH_OK) die(qh_last_error());
      piece pc;
      qh_correct_stats st;
      if (qh_corrector_fetch(ec, &pc.fa, &pc.nfa, &pc.lg, &pc.nlg, &st) != QH_OK) die(qh_last_error());
      out.push_back(pc);
      add_stats(st);
    }
    const double t2 = secs();
    join_writer();  // chunk i-1 is out: its buffers are the ones the next fetch reuses
    writer = std::thread([out, fa_fd, log_fd, &fa_name, &log_name]() {
      for (const piece& pc : out) {
        write_all(fa_fd, pc.fa, pc.nfa, fa_name);
        write_all(log_fd, pc.lg, pc.nlg, log_name);
      }
    });
    t_correct += t2 - t1;
    t_write += secs() - t2;
    return used;
  };
  // The plain loop: read a chunk, correct it, carry the partial record at its end over. `have` bytes of the file are
  // already in buf (the pipe loop below hands over to it when a record outgrows its buffers).
  auto run_serial = [&](int fd, bool regular, off_t fpos, size_t have, bool eof, const char* path) {
    while (!eof || have) {
      if (!eof && have < cap) {
        const double t0 = secs();
        const size_t want = cap - have, r = read_some(fd, regular, &fpos, buf + have, want);
        t_read += secs() - t0;
        have += r;
        if (r < want) eof = true;
      }
      if (have == 0) break;
      const size_t used = process(buf, have, eof);
      if (used == 0) {
        if (eof) die(std::string("Truncated fastq file '") + path + "'");
        // a record longer than the buffer: grow it
        char* nb = nullptr;
        const size_t ncap = cap * 2;
        bool npinned = qh_host_alloc((void**)&nb, ncap) == QH_OK;
        if (!npinned) nb = (char*)malloc(ncap);
        if (!nb) die("out of memory for the chunk buffer");
        memcpy(nb, buf, have);
        if (pinned) qh_host_free(buf); else free(buf);
        buf = nb; cap = ncap; pinned = npinned;
        continue;
      }
      memmove(buf, buf + used, have - used);
      have -= used;
    }
  };
  // A pipe (the paired-end plumbing, zcat | ...) delivers at the pace of its writer: a second thread fills the next
  // chunk's buffer while this one corrects. Two buffers with `head` spare bytes in front, where the partial record of
  // the previous chunk is put before the new bytes; a record that outgrows that goes back to the plain loop.
  auto run_piped = [&](int fd, const char* path) {
    (void)fcntl(fd, F_SETPIPE_SZ, 1 << 20);  // a pipe: fewer, larger transfers (no-op on anything else)
    const size_t head = cap / 4 < ((size_t)16 << 20) ? cap / 4 : ((size_t)16 << 20);
    char* pb[2] = {nullptr, nullptr};
    bool ppinned[2] = {false, false};
    for (int i = 0; i < 2; ++i) {
      ppinned[i] = qh_host_alloc((void**)&pb[i], head + cap) == QH_OK;
      if (!ppinned[i]) pb[i] = (char*)malloc(head + cap);
      if (!pb[i]) die("out of memory for the chunk buffer");
    }
    struct fill_t { size_t n; bool eof; };
    fill_t got[2] = {{0, false}, {0, false}};
    std::thread reader;
    auto start_read = [&](int j) {
      reader = std::thread([&, j]() {
        off_t unused = 0;
        const size_t r = read_some(fd, false, &unused, pb[j] + head, cap);
        got[j] = fill_t{r, r < cap};
      });
    };
    start_read(0);
    size_t tail = 0;
    int j = 0;
    bool handed_over = false;
    for (;;) {
      const double t0 = secs();
      reader.join();
      t_read += secs() - t0;
      const fill_t f = got[j];
      char* chunk = pb[j] + head - tail;
      size_t have = tail + f.n;
      if (!f.eof) start_read(j ^ 1);
      if (have == 0) break;
      size_t used = process(chunk, have, f.eof);
      if (f.eof) {
        // what a final chunk leaves is an error of the input; the corrector names it when asked about the rest alone
        while (used < have) {
          if (used == 0) die(std::string("Truncated fastq file '") + path + "'");
          chunk += used; have -= used;
          used = process(chunk, have, true);
        }
        break;
      }
      tail = have - used;
      if (used == 0 || tail > head) {
        // a record longer than a buffer's spare room: everything read so far into one growing buffer, plain loop from here
        reader.join();
        const fill_t g = got[j ^ 1];
        const size_t need = tail + g.n;
        if (need > cap) {
          char* nb = nullptr;
          const size_t ncap = need > cap * 2 ? need : cap * 2;
          bool npinned = qh_host_alloc((void**)&nb, ncap) == QH_OK;
          if (!npinned) nb = (char*)malloc(ncap);
          if (!nb) die("out of memory for the chunk buffer");
          if (pinned) qh_host_free(buf); else free(buf);
          buf = nb; cap = ncap; pinned = npinned;
        }
        memcpy(buf, chunk + used, tail);
        memcpy(buf + tail, pb[j ^ 1] + head, g.n);
        for (int i = 0; i < 2; ++i) { if (ppinned[i]) qh_host_free(pb[i]); else free(pb[i]); }
        run_serial(fd, false, 0, need, g.eof, path);
        handed_over = true;
        break;
      }
      memcpy(pb[j ^ 1] + head - tail, chunk + used, tail);  // in front of what the reader is bringing
      j ^= 1;
    }
    if (!handed_over)
      for (int i = 0; i < 2; ++i) { if (ppinned[i]) qh_host_free(pb[i]); else free(pb[i]); }
  };
  for (const char* path : args.sequence) {
    const int fd = !strcmp(path, "-") || !strcmp(path, "/dev/fd/0") ? 0 : open(path, O_RDONLY);
    if (fd < 0) die(std::string("Can't open file '") + path + "'");
    struct stat fst;
    const bool regular = fd != 0 && fstat(fd, &fst) == 0 && S_ISREG(fst.st_mode);
    if (regular || getenv("QUORUM_HIP_NO_READ_AHEAD")) run_serial(fd, regular, 0, 0, false, path);
    else run_piped(fd, path);
    if (fd != 0) close(fd);
  }
  { const double t0 = secs(); join_writer(); t_write += secs() - t0; }
  if (args.verbose) fprintf(stderr, "[%.3f] seconds: read %.3f  correct+copy %.3f  write %.3f\n", secs(), t_read, t_correct, t_write);
  if (args.verbose)
    fprintf(stderr, "[%.3f] Done: reads %llu corrected %llu skipped %llu substitutions %llu truncations %llu bases %llu -> %llu\n",
            secs(), (unsigned long long)total.reads, (unsigned long long)total.corrected, (unsigned long long)total.skipped,
            (unsigned long long)total.substitutions, (unsigned long long)total.truncations,
            (unsigned long long)total.bases_in, (unsigned long long)total.bases_out);
  if (fa_fd > 2) close(fa_fd);
  if (log_fd > 2) close(log_fd);
  if (pinned) qh_host_free(buf); else free(buf);
  for (auto& r : rep) {
    qh_corrector_destroy(r.ec);
    if (r.contaminant) qh_query_close(r.contaminant);
    qh_query_close(r.db);
  }
  return 0;
}
