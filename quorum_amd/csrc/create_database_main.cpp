// quorum_create_database -- drop-in command line of Quorum's database builder on
// top of libquorum_hip.
//
// Mirrors src/create_database.cc:98-129 (main) and the switch set of
// src/create_database_cmdline.yaggo:7-33: -s/--size (uint64 with SI suffix,
// required), -m/--mer, -b/--bits (required), -q/--min-qual-value,
// -Q/--min-qual-char (conflict), -t/--threads (=1), -o/--output
// (="combined_database"), -p/--reprobe (=126), reads... (at least 1).
// Not in upstream: --gpu N selects the HIP device; --gpus N builds the table hash-sharded over devices 0..N-1 (N a power
// of two; --devices a,b,... names them) in this one process and serialises the shards into the one output file; -t is
// accepted and ignored (the GPUs supply the parallelism).
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <chrono>
#include <thread>
#include <unistd.h>
#include <fcntl.h>
#include <sys/stat.h>

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
    "     --gpus=int                           Build hash-sharded over this many GPUs (1)\n"
    "     --devices=list                       The HIP devices of the shards, comma separated\n"
    "     --jf-layout                          Keys image in the (unverified) Jellyfish layout; 1 GPU\n"
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
  int gpus = 1;
  std::vector<int> devices;
  bool jf_layout = false;
  std::vector<const char*> reads;
};

static void parse(int argc, char** argv, args_t& a) {
  struct opt { char s; const char* l; };
  static const opt opts[] = {{'s', "size"}, {'m', "mer"}, {'b', "bits"}, {'q', "min-qual-value"},
                             {'Q', "min-qual-char"}, {'t', "threads"}, {'o', "output"}, {'p', "reprobe"},
                             {'G', "gpu"}, {'N', "gpus"}, {'D', "devices"}};
  bool only_pos = false;
  for (int i = 1; i < argc; ++i) {
    const char* s = argv[i];
    if (only_pos || s[0] != '-' || s[1] == 0) { a.reads.push_back(s); continue; }
    if (!strcmp(s, "--")) { only_pos = true; continue; }
    if (!strcmp(s, "-h") || !strcmp(s, "--help")) { printf("%s\n\n%s\n", kUsage, kHelp); exit(0); }
    if (!strcmp(s, "-U") || !strcmp(s, "--usage")) { printf("%s\n", kUsage); exit(0); }
    if (!strcmp(s, "-V") || !strcmp(s, "--version")) { printf("%s\n", qh_version()); exit(0); }
    if (!strcmp(s, "--jf-layout")) { a.jf_layout = true; continue; }
    char key = 0;
    const char* val = nullptr;
    if (s[1] == '-') {
      const char* eq = strchr(s + 2, '=');
      const std::string name(s + 2, eq ? (size_t)(eq - s - 2) : strlen(s + 2));
      for (auto& o : opts) if (name == o.l) key = o.s;
      if (!key) error(std::string("unrecognized option '") + s + "'");
      val = eq ? eq + 1 : nullptr;
    } else {
      for (auto& o : opts) if (s[1] == o.s && o.s != 'G' && o.s != 'N' && o.s != 'D') key = o.s;
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
      case 'N': a.gpus = (int)conv_uint(val, false, "int", 256); break;
      case 'D': {
        a.devices.clear();
        std::string item;
        for (const char* c = val;; ++c) {
          if (*c == ',' || *c == 0) { a.devices.push_back((int)conv_uint(item.c_str(), false, "int", 1024)); item.clear(); if (!*c) break; }
          else item.push_back(*c);
        }
        break;
      }
    }
  }
  if (!a.size_given) error("[-s, --size=uint64] required switch");
  if (!a.mer_given) error("[-m, --mer=uint32] required switch");
  if (!a.bits_given) error("[-b, --bits=uint32] required switch");
  if (a.min_qual_value_given && a.min_qual_char_given)
    error("Switches [-Q, --min-qual-char=string] and [-q, --min-qual-value=uint32] are mutually exclusive");
  if (a.reads.empty()) error("Requires at least 1 argument.");
  if (!a.devices.empty()) a.gpus = (int)a.devices.size();
  if (a.gpus < 1 || (a.gpus & (a.gpus - 1))) error("The number of GPUs must be a power of two");
  if (a.jf_layout && a.gpus > 1) error("Switch [--jf-layout] writes from one GPU: not with --gpus / --devices");
}

[[noreturn]] static void die(const char* what, int rc) {
  fprintf(stderr, "%s: %s\n", what, qh_last_error());
  // upstream: an uncaught runtime_error("Hash is full") in a worker -> abort; the
  // driver only checks for a non-zero status (src/quorum.in:155-158)
  exit(rc == QH_EFULL ? 134 : EXIT_FAILURE);
}

// Chunk buffer in page-locked memory (plain malloc when pinning fails)
struct chunk_buf {
  char* p = nullptr;
  size_t cap = 0;
  bool pinned = false;
  void reserve(size_t n) {
    if (n <= cap) return;
    char* np = nullptr;
    bool npinned = qh_host_alloc((void**)&np, n) == QH_OK;
    if (!npinned) np = (char*)malloc(n);
    if (!np) { fprintf(stderr, "out of memory for a %zu-byte chunk\n", n); exit(EXIT_FAILURE); }
    if (p) memcpy(np, p, cap);
    release();
    p = np; cap = n; pinned = npinned;
  }
  void release() {
    if (!p) return;
    if (pinned) qh_host_free(p); else free(p);
    p = nullptr; cap = 0;
  }
  ~chunk_buf() { release(); }
};

// Fills dst[0, n) from the file: a regular file is read by a few threads at once (pread of
// disjoint slices: the page-cache copy is what bounds the tool), anything else sequentially.
static size_t read_some(int fd, bool regular, off_t* pos, char* dst, size_t n) {
  if (!regular) {
    size_t got = 0;
    while (got < n) {
      ssize_t r = read(fd, dst + got, n - got);
      if (r < 0) { if (errno == EINTR) continue; perror("read"); exit(EXIT_FAILURE); }
      if (r == 0) break;
      got += (size_t)r;
    }
    return got;
  }
  struct stat st;
  if (fstat(fd, &st) != 0) { perror("fstat"); exit(EXIT_FAILURE); }
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
  if (failed) { fprintf(stderr, "read error\n"); exit(EXIT_FAILURE); }
  *pos += (off_t)n;
  return n;
}

// Feeds one read file. Fast path: cut the stream into chunks of whole 4-line
// groups and let the GPU index + validate them; the first chunk the GPU rejects
// switches the rest of the file to the sequential host parser (Jellyfish rules).
static void feed_file(qh_build* h, const char* path, size_t chunk_bytes) {
  const int fd = !strcmp(path, "-") ? 0 : open(path, O_RDONLY);
  if (fd < 0) { fprintf(stderr, "Can't open file '%s'\n", path); exit(EXIT_FAILURE); }
  struct stat st;
  const bool regular = fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
  if (regular && (size_t)st.st_size + 16 < chunk_bytes) chunk_bytes = (size_t)st.st_size + 16;
  off_t pos = 0;
  chunk_buf buf;
  buf.reserve(chunk_bytes);
  size_t have = 0;
  bool fast = true, eof = false, any = false;
  while (!eof || have) {
    if (!eof && have < buf.cap) {
      const size_t want = buf.cap - have, r = read_some(fd, regular, &pos, buf.p + have, want);
      have += r;
      if (r < want) eof = true;
    }
    if (have == 0) break;
    if (!any && buf.p[0] != '@' ) {
      fprintf(stderr, "Unsupported format in file '%s' (create_database needs FASTQ: it reads a quality for every base)\n", path);
      exit(EXIT_FAILURE);
    }
    any = true;
    size_t used = 0;
    if (fast) {
      used = qh_fastq_4line_prefix(buf.p, have, eof);
      int rc = used ? qh_build_add_fastq(h, buf.p, used) : QH_OK;
      if (rc == QH_OK && used) rc = qh_build_sync(h);
      if (rc == QH_EFORMAT || (used == 0 && (eof || have == buf.cap))) { fast = false; used = 0; }
      else if (rc != QH_OK) die("create_database", rc);
    }
    if (!fast) {
      int rc = qh_build_add_fastq_host_parsed(h, buf.p, have, eof, &used);
      if (rc == QH_OK) rc = qh_build_sync(h);
      if (rc != QH_OK) die("create_database", rc);
      if (used == 0 && !eof && have == buf.cap) buf.reserve(buf.cap * 2);  // a record longer than the chunk
    }
    memmove(buf.p, buf.p + used, have - used);
    have -= used;
    if (eof && used == 0 && have) {  // cannot happen: final chunks parse completely or fail
      fprintf(stderr, "Truncated fastq file '%s'\n", path);
      exit(EXIT_FAILURE);
    }
  }
  if (fd != 0) close(fd);
}

// The same for the sharded build: the chunk is handed to qh_multi_add_fastq, which cuts it into one part per GPU.
static void feed_file_multi(qh_multi* m, const char* path, size_t chunk_bytes) {
  const int fd = !strcmp(path, "-") ? 0 : open(path, O_RDONLY);
  if (fd < 0) { fprintf(stderr, "Can't open file '%s'\n", path); exit(EXIT_FAILURE); }
  struct stat st;
  const bool regular = fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
  if (regular && (size_t)st.st_size + 16 < chunk_bytes) chunk_bytes = (size_t)st.st_size + 16;
  off_t pos = 0;
  chunk_buf buf;
  buf.reserve(chunk_bytes);
  size_t have = 0;
  bool eof = false, any = false;
  while (!eof || have) {
    if (!eof && have < buf.cap) {
      const size_t want = buf.cap - have, r = read_some(fd, regular, &pos, buf.p + have, want);
      have += r;
      if (r < want) eof = true;
    }
    if (have == 0) break;
    if (!any && buf.p[0] != '@') {
      fprintf(stderr, "Unsupported format in file '%s' (create_database needs FASTQ: it reads a quality for every base)\n", path);
      exit(EXIT_FAILURE);
    }
    any = true;
    size_t used = 0;
    const int rc = qh_multi_add_fastq(m, buf.p, have, eof, &used);
    if (rc != QH_OK) die("create_database", rc);
    if (used == 0 && !eof && have == buf.cap) buf.reserve(buf.cap * 2);  // a record longer than the chunk
    memmove(buf.p, buf.p + used, have - used);
    have -= used;
    if (eof && used == 0 && have) { fprintf(stderr, "Truncated fastq file '%s'\n", path); exit(EXIT_FAILURE); }
  }
  if (fd != 0) close(fd);
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

  const auto t_start = std::chrono::steady_clock::now();
  auto secs = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
  const char* cb = getenv("QUORUM_HIP_CHUNK_MB");
  size_t chunk = (cb && atol(cb) > 0 ? (size_t)atol(cb) : 256) << 20;
  if (!(cb && atol(cb) > 0)) {
    // A chunk that goes through the partitioned build streams every window of the table once: the chunk should bring
    // a GPU about half as many k-mers as its table has slots (about 0.4 k-mers per byte of FASTQ), within the 4 GB a
    // chunk may have. 2^26 slots stay at 256 MiB; the 2^32-slot table of a human genome reads 3.75 GiB at a time.
    uint64_t slots = 1;
    while (slots < args.size && slots < (1ull << 62)) slots <<= 1;
    const uint64_t per_gpu = slots / (uint64_t)args.gpus, want = per_gpu + per_gpu / 4;
    const uint64_t most = (uint64_t)3840 << 20;
    if (want > chunk) chunk = (size_t)(want < most ? want : most);
  }
  if (args.gpus > 1) {  // the table hash-sharded over several GPUs, one process (src/create_database.cc:117-125)
    qh_multi* m = nullptr;
    int mrc = qh_multi_create(&m, (int)args.mer, (int)args.bits, args.size, (int)qual_thresh, args.reprobe, args.gpus,
                              args.devices.empty() ? nullptr : args.devices.data());
    if (mrc != QH_OK) error(qh_last_error());
    const double t_init = secs();
    for (const char* path : args.reads) feed_file_multi(m, path, chunk * (size_t)args.gpus);
    qh_build_stats st;
    if ((mrc = qh_multi_finish(m, &st)) != QH_OK) die("create_database", mrc);
    const double t_feed = secs();
    if ((mrc = qh_multi_write(m, args.output, argc, argv)) != QH_OK) die("create_database", mrc);
    if (getenv("QUORUM_HIP_VERBOSE")) {
      int uses_rccl = 0;
      uint64_t rounds = 0, fb = 0, retries = 0;
      qh_multi_info(m, nullptr, &uses_rccl, &rounds, &fb, &retries);
      fprintf(stderr, "seconds: devices+tables %.3f  read+build %.3f  write %.3f\n", t_init, t_feed - t_init, secs() - t_feed);
      fprintf(stderr, "gpus %d  exchange %s  rounds %llu  host-parsed chunks %llu  bucket retries %llu\n", args.gpus,
              uses_rccl ? "rccl" : "peer copies", (unsigned long long)rounds, (unsigned long long)fb, (unsigned long long)retries);
      fprintf(stderr, "reads %llu  k-mers %llu  high-quality %llu  distinct %llu  slots %llu  doublings %llu\n",
              (unsigned long long)st.records, (unsigned long long)st.instances, (unsigned long long)st.hq_instances,
              (unsigned long long)st.distinct, (unsigned long long)st.slots, (unsigned long long)st.grows);
    }
    qh_multi_destroy(m);
    return 0;
  }
  qh_build* h = nullptr;
  int rc = qh_build_create(&h, (int)args.mer, (int)args.bits, args.size, (int)qual_thresh, args.reprobe, args.gpu, 0, 0);
  if (rc != QH_OK) error(qh_last_error());

  const double t_init = secs();
  for (const char* path : args.reads) feed_file(h, path, chunk);

  qh_build_stats st;
  if ((rc = qh_build_finish(h, &st)) != QH_OK) die("create_database", rc);
  const double t_feed = secs();
  rc = args.jf_layout ? qh_build_write_jf(h, args.output, argc, argv) : qh_build_write(h, args.output, argc, argv);
  if (rc != QH_OK) die("create_database", rc);
  if (getenv("QUORUM_HIP_VERBOSE"))
    fprintf(stderr, "seconds: device+table %.3f  read+build %.3f  write %.3f\n", t_init, t_feed - t_init, secs() - t_feed);
  if (getenv("QUORUM_HIP_VERBOSE"))
    fprintf(stderr, "reads %llu  k-mers %llu  high-quality %llu  distinct %llu  slots %llu  doublings %llu\n",
            (unsigned long long)st.records, (unsigned long long)st.instances, (unsigned long long)st.hq_instances,
            (unsigned long long)st.distinct, (unsigned long long)st.slots, (unsigned long long)st.grows);
  qh_build_destroy(h);
  return 0;
}
