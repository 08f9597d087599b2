/*
 * quorum_oracle.c -- CPU restatement of Quorum's k-mer database build and lookup.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (quorum_amd/, include/,
 * the libquorum_hip C-ABI, the CLI) may include, link, import or execute this
 * file.  It is used by tests/, by __graft_entry__.smoke() and by bench.py's
 * cpu_baseline leg, and there only as the checker / the reported CPU baseline.
 *
 * What it restates (paths relative to the upstream Quorum tree):
 *   - per-base extraction loop            src/create_database.cc:56-95
 *   - hash_with_quality ctor / add        src/mer_database.hpp:78-87, 94-113
 *   - table doubling (serial form)        src/mer_database.hpp:137-187
 *   - database_query operator[], get_val  src/mer_database.hpp:284-299
 *   - get_best_alternatives               src/mer_database.hpp:302-329
 *     (forward_mer / backward_mer "base 0" meaning: src/kmer.hpp:47-50,96-103)
 *   - Poisson-cutoff table reduction      src/error_correct_reads.cc:650-668
 *   - histogram                           src/histo_mer_database.cc:13-26
 *
 * The arithmetic below the Quorum sources lives in Jellyfish 2.x (pkg-config
 * module jellyfish-2.0, version unpinned by configure.ac:28-29), which is NOT
 * vendored in the reference tree and is not installed here.  Its published
 * algorithm is restated: mer_dna 2-bit codec (A,C,G,T -> 0..3, base 0 in the two
 * low bits), large_hash::array (GF(2) matrix position hash, quadratic reprobe
 * i(i+1)/2, claim by word CAS), atomic_bits_array (bits+1 wide fields,
 * floor(64/(bits+1)) per word, whole-word CAS), whole_sequence_parser FASTQ rules.
 *
 * PARITY PIN.  The update rule / lookup semantics are pinned by the reference's
 * own known answers (unit_tests/test_mer_database.cc:115-120: the six classes
 * hq2,hq1,lqhq,hqlq,lq1,lq2 -> (2,1),(1,1),(1,1),(1,1),(1,0),(2,0)) plus the
 * header/file-length invariants (:102-106,:113-114); tests/test_oracle.py runs
 * them.  The reference holds no golden vector for FASTQ extraction, quality
 * thresholding or canonicalisation: for those this oracle is a line-by-line
 * restatement of create_database.cc:64-92 and is "parity unpinned" beyond the
 * hand-derived cases of SURVEY.md Appendix A (tests/golden/).
 * oracle/_ref (a build of the real reference) is NOT possible: the sources need
 * Jellyfish headers and a yaggo-generated header that this image lacks.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

/* ------------------------------------------------------------------ mer_dna */

/* jellyfish::mer_dna::code: A/a C/c G/g T/t -> 0..3; anything else negative
 * (create_database.cc:73-74 only tests "< 0"). */
static int8_t g_code[256];
static pthread_once_t g_code_once = PTHREAD_ONCE_INIT;
static void init_code(void) {
  memset(g_code, -3, sizeof(g_code));
  g_code['\n'] = -2;
  g_code['N'] = g_code['n'] = -1;
  g_code['A'] = g_code['a'] = 0;
  g_code['C'] = g_code['c'] = 1;
  g_code['G'] = g_code['g'] = 2;
  g_code['T'] = g_code['t'] = 3;
}

int qo_code(int c) {
  pthread_once(&g_code_once, init_code);
  return g_code[(unsigned char)c];
}

static inline uint64_t mer_mask(int k) { return k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1); }

/* reverse complement of a k-mer held in the low 2k bits */
uint64_t qo_revcomp(uint64_t m, int k) {
  uint64_t r = 0;
  for (int i = 0; i < k; ++i) {
    r = (r << 2) | (3 - (m & 3));
    m >>= 2;
  }
  return r;
}

/* mer_dna::canonical(): numeric min of the mer and its reverse complement
 * (create_database.cc:86 "m < rm ? m : rm"). */
uint64_t qo_canonical(uint64_t m, int k) {
  uint64_t r = qo_revcomp(m, k);
  return m < r ? m : r;
}

/* string (first char = base k-1) -> mer; returns -1 on a non-ACGT char */
int qo_mer_from_str(const char* s, int k, uint64_t* out) {
  uint64_t m = 0;
  for (int i = 0; i < k; ++i) {
    int c = qo_code(s[i]);
    if (c < 0) return -1;
    m = (m << 2) | (uint64_t)c;
  }
  *out = m;
  return 0;
}

/* ------------------------------------------------- extraction of one read */

/* create_database.cc:64-92 for ONE read.  m/rm are the caller's rolling
 * registers (never cleared between reads or at an N, exactly as upstream :57).
 * Emits (canonical mer, hq) pairs into mers[]/hq[] (capacity >= len) and returns
 * how many.  thresh is compared as a signed char (:48,:81). */
size_t qo_extract_read(const char* seq, const char* qual, size_t len, int k, int thresh,
                       uint64_t* m_io, uint64_t* rm_io, uint64_t* mers, uint8_t* hq) {
  pthread_once(&g_code_once, init_code);
  const uint64_t mask = mer_mask(k);
  const signed char qt = (signed char)thresh;
  uint64_t m = *m_io, rm = *rm_io;
  unsigned low_len = 0, high_len = 0;
  size_t n = 0;
  for (size_t i = 0; i < len; ++i) {
    int code = g_code[(unsigned char)seq[i]];
    if (code < 0) {                       /* :74-77 */
      high_len = low_len = 0;
      continue;
    }
    m = ((m << 2) | (uint64_t)code) & mask;                       /* shift_left  :78 */
    rm = (rm >> 2) | ((uint64_t)(3 - code) << (2 * (k - 1)));     /* shift_right :79 */
    ++low_len;
    if ((signed char)qual[i] >= qt) ++high_len; else high_len = 0; /* :81-84 */
    if (low_len >= (unsigned)k) {                                  /* :85-86 */
      mers[n] = m < rm ? m : rm;
      hq[n] = high_len >= (unsigned)k;
      ++n;
    }
  }
  *m_io = m; *rm_io = rm;
  return n;
}

/* ------------------------------------------------------------ FASTQ parser */

/* jellyfish::whole_sequence_parser::read_fastq restated for the strictly
 * sequential case.  A record: skip 1 char ('@'), header = rest of line; sequence
 * lines are appended until a line starting with '+' (or EOF -> "Truncated");
 * that line is skipped; quality lines are appended until qual.size() >=
 * seq.size(); sizes must then be equal and the next char must be EOF or '@'.
 * Records are returned as (seq_off, seq_len, qual_off) when the sequence and
 * quality are each a single line; multi-line records are re-flowed into the
 * caller's scratch buffer.  Error codes: -1 truncated, -2 wrong number of
 * quals, -3 header missing, -4 unsupported format (first char not '@'),
 * -5 out of capacity. */
typedef struct { uint64_t seq_off, qual_off; uint32_t len; uint32_t in_scratch; } qo_rec;

static size_t line_end(const char* buf, size_t len, size_t p) {
  const char* e = (const char*)memchr(buf + p, '\n', len - p);
  return e ? (size_t)(e - buf) : len;
}

int64_t qo_parse_fastq_hdr(const char* buf, size_t len, qo_rec* recs, size_t cap,
                           char* scratch, size_t scratch_cap, uint64_t* hdr_off, uint32_t* hdr_len) {
  size_t p = 0, n = 0, sp = 0;
  if (len == 0) return 0;
  if (buf[0] != '@') return -4;
  while (p < len) {
    if (n >= cap) return -5;
    /* is.ignore(1); getline(header) */
    size_t e = line_end(buf, len, p + 1 <= len ? p + 1 : len);
    if (hdr_off) { hdr_off[n] = p + 1 <= len ? p + 1 : len; hdr_len[n] = (uint32_t)(e - hdr_off[n]); }
    p = e < len ? e + 1 : len;
    /* sequence lines */
    size_t s0 = p, slen = 0, nlines = 0, first_len = 0;
    size_t seq_scratch = sp;
    while (p < len && buf[p] != '+') {
      e = line_end(buf, len, p);
      size_t l = e - p;
      if (nlines == 0) first_len = l;
      if (nlines == 1) {           /* second line: start re-flowing */
        if (sp + first_len > scratch_cap) return -5;
        memcpy(scratch + sp, buf + s0, first_len); sp += first_len;
      }
      if (nlines >= 1) {
        if (sp + l > scratch_cap) return -5;
        memcpy(scratch + sp, buf + p, l); sp += l;
      }
      slen += l; ++nlines;
      p = e < len ? e + 1 : len;
    }
    if (p >= len) return -1;       /* "Truncated fastq file" */
    /* skip the '+' line */
    e = line_end(buf, len, p);
    int plus_hit_eof = e >= len;
    p = e < len ? e + 1 : len;
    /* quality lines until qual.size() >= seq.size() (and stream still good) */
    size_t q0 = p, qlen = 0, qlines = 0, qfirst = 0;
    size_t qual_scratch = sp;
    int good = !plus_hit_eof;
    while (qlen < slen && good) {
      if (p >= len) { good = 0; break; }
      e = line_end(buf, len, p);
      size_t l = e - p;
      if (qlines == 0) qfirst = l;
      if (qlines == 1) {
        if (sp + qfirst > scratch_cap) return -5;
        memcpy(scratch + sp, buf + q0, qfirst); sp += qfirst;
      }
      if (qlines >= 1) {
        if (sp + l > scratch_cap) return -5;
        memcpy(scratch + sp, buf + p, l); sp += l;
      }
      qlen += l; ++qlines;
      if (e >= len) good = 0;      /* getline hit EOF: eofbit set */
      p = e < len ? e + 1 : len;
    }
    if (qlen != slen) return -2;   /* "wrong number of quals" */
    if (p < len && buf[p] != '@') return -3; /* "header missing" */
    qo_rec* r = &recs[n++];
    r->len = (uint32_t)slen;
    if (nlines <= 1 && qlines <= 1) {
      r->seq_off = s0; r->qual_off = q0; r->in_scratch = 0;
    } else {
      /* at least one side was multi-line: make both live in scratch */
      if (nlines <= 1) {
        if (sp + slen > scratch_cap) return -5;
        seq_scratch = sp; memcpy(scratch + sp, buf + s0, slen); sp += slen;
      }
      if (qlines <= 1) {
        if (sp + qlen > scratch_cap) return -5;
        qual_scratch = sp; memcpy(scratch + sp, buf + q0, qlen); sp += qlen;
      }
      r->seq_off = seq_scratch; r->qual_off = qual_scratch; r->in_scratch = 1;
    }
  }
  return (int64_t)n;
}

int64_t qo_parse_fastq(const char* buf, size_t len, qo_rec* recs, size_t cap,
                       char* scratch, size_t scratch_cap) {
  return qo_parse_fastq_hdr(buf, len, recs, cap, scratch, scratch_cap, NULL, NULL);
}

/* ------------------------------------------------------- hash_with_quality */

/* large_hash::array restated: size rounded up to a power of two, position =
 * GF(2) matrix times key, reprobe offsets i(i+1)/2, claim by CAS.  The full key
 * (+1, 0 = empty) is stored in a 64-bit word instead of Jellyfish's bit-packed
 * remainder field: slot layout is not part of the content contract.
 * atomic_bits_array restated: (bits+1)-wide fields, 64/(bits+1) per word. */
typedef struct qo_db {
  int k, bits, vbits, vpw;
  unsigned lsize, max_reprobe, reprobe_limit;
  uint64_t size, max_val, vmask;
  uint64_t* keys;           /* key+1, 0 = empty */
  uint64_t* vals;           /* packed value fields */
  uint64_t lut[8][256];     /* byte-sliced GF(2) matrix (64 columns) */
  uint64_t seed;
} qo_db;

static uint64_t splitmix(uint64_t* s) {
  uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

static void make_matrix(qo_db* db) {
  uint64_t col[64];
  for (int i = 0; i < 64; ++i) col[i] = splitmix(&db->seed);
  for (int b = 0; b < 8; ++b)
    for (int v = 0; v < 256; ++v) {
      uint64_t x = 0;
      for (int j = 0; j < 8; ++j)
        if (v & (1 << j)) x ^= col[b * 8 + j];
      db->lut[b][v] = x;
    }
}

static inline uint64_t hash_pos(const qo_db* db, uint64_t key) {
  uint64_t x = 0;
  for (int b = 0; b < 8; ++b) x ^= db->lut[b][(key >> (8 * b)) & 0xff];
  return x & (db->size - 1);
}

static int alloc_arrays(qo_db* db) {
  db->keys = (uint64_t*)calloc(db->size, sizeof(uint64_t));
  db->vals = (uint64_t*)calloc((db->size + db->vpw - 1) / db->vpw, sizeof(uint64_t));
  if (!db->keys || !db->vals) { free(db->keys); free(db->vals); db->keys = db->vals = 0; return -1; }
  return 0;
}

static void set_geometry(qo_db* db, uint64_t size, unsigned reprobe_limit) {
  unsigned l = 0;
  while ((1ull << l) < size) ++l;
  db->lsize = l; db->size = 1ull << l;
  /* quadratic_reprobes: lower the limit until reprobes[limit] < size */
  unsigned r = reprobe_limit;
  while (r > 0 && (uint64_t)r * (r + 1) / 2 >= db->size) --r;
  db->max_reprobe = r;
}

qo_db* qo_db_create(uint64_t size, int k, int bits, int reprobe_limit) {
  if (k < 1 || k > 31 || bits < 1 || bits > 63 || size == 0) return 0;
  qo_db* db = (qo_db*)calloc(1, sizeof(qo_db));
  if (!db) return 0;
  db->k = k; db->bits = bits; db->vbits = bits + 1; db->vpw = 64 / db->vbits;
  db->max_val = ~0ull >> (64 - bits);                        /* mer_database.hpp:83 */
  db->vmask = db->vbits >= 64 ? ~0ull : ((1ull << db->vbits) - 1);
  db->seed = 0x51f15eedull;
  db->reprobe_limit = (unsigned)reprobe_limit;
  set_geometry(db, size, (unsigned)reprobe_limit);
  make_matrix(db);
  if (alloc_arrays(db)) { free(db); return 0; }
  return db;
}

void qo_db_destroy(qo_db* db) {
  if (!db) return;
  free(db->keys); free(db->vals); free(db);
}

/* array::set : returns slot id or -1 when the reprobe limit is exhausted */
static inline int64_t key_set(qo_db* db, uint64_t key) {
  const uint64_t want = key + 1, msk = db->size - 1;
  uint64_t pos = hash_pos(db, key);
  for (unsigned i = 0; i <= db->max_reprobe; ++i) {
    uint64_t id = (pos + (uint64_t)i * (i + 1) / 2) & msk;
    uint64_t cur = __atomic_load_n(&db->keys[id], __ATOMIC_RELAXED);
    if (cur == 0) {
      if (__atomic_compare_exchange_n(&db->keys[id], &cur, want, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED))
        return (int64_t)id;
    }
    if (cur == want) return (int64_t)id;
  }
  return -1;
}

static inline int64_t key_get(const qo_db* db, uint64_t key) {
  const uint64_t want = key + 1, msk = db->size - 1;
  uint64_t pos = hash_pos(db, key);
  for (unsigned i = 0; i <= db->max_reprobe; ++i) {
    uint64_t id = (pos + (uint64_t)i * (i + 1) / 2) & msk;
    uint64_t cur = db->keys[id];
    if (cur == want) return (int64_t)id;
    if (cur == 0) return -1;
  }
  return -1;
}

static inline uint64_t val_get(const qo_db* db, uint64_t id) {
  return (db->vals[id / db->vpw] >> ((id % db->vpw) * db->vbits)) & db->vmask;
}

/* the CAS loop of mer_database.hpp:102-111, on the whole word like
 * atomic_bits_array::set (on failure nval is reloaded from the fresh word). */
static inline void val_update(qo_db* db, uint64_t id, unsigned quality) {
  uint64_t* w = &db->vals[id / db->vpw];
  const unsigned sh = (unsigned)((id % db->vpw) * db->vbits);
  uint64_t word = __atomic_load_n(w, __ATOMIC_RELAXED);
  for (;;) {
    uint64_t nval = (word >> sh) & db->vmask;
    if ((nval & 1) < quality) nval = 3;
    else if ((nval >> 1) == db->max_val || (nval & 1) > quality) return;
    else nval += 2;
    uint64_t nword = (word & ~(db->vmask << sh)) | (nval << sh);
    if (__atomic_compare_exchange_n(w, &word, nword, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return;
  }
}

/* serial form of handle_full_ary (mer_database.hpp:137-187): 2x table, fresh
 * matrix, re-insert keys, copy values. Returns -1 when allocation fails (FULL). */
static int grow(qo_db* db) {
  const qo_db old = *db;
  for (uint64_t nsize = old.size * 2; ; nsize *= 2) {
    set_geometry(db, nsize, db->reprobe_limit);
    make_matrix(db);
    if (alloc_arrays(db)) { *db = old; return -1; }
    int ok = 1;
    for (uint64_t id = 0; id < old.size && ok; ++id) {
      if (!old.keys[id]) continue;
      int64_t nid = key_set(db, old.keys[id] - 1);
      if (nid < 0) { ok = 0; break; }   /* reprobe limit hit again: go bigger */
      uint64_t v = val_get(&old, id);
      db->vals[nid / db->vpw] |= v << ((nid % db->vpw) * db->vbits);
    }
    if (ok) break;
    free(db->keys); free(db->vals);
  }
  free(old.keys); free(old.vals);
  return 0;
}

/* hash_with_quality::add (serial callers may trigger growth). 0 ok, -1 full */
int qo_db_add(qo_db* db, uint64_t mer, unsigned quality) {
  int64_t id;
  while ((id = key_set(db, mer)) < 0)
    if (grow(db)) return -1;
  val_update(db, (uint64_t)id, quality ? 1 : 0);
  return 0;
}

/* concurrent form: no growth; -1 = "Hash is full" */
static inline int add_nogrow(qo_db* db, uint64_t mer, unsigned quality) {
  int64_t id = key_set(db, mer);
  if (id < 0) return -1;
  val_update(db, (uint64_t)id, quality);
  return 0;
}

int64_t qo_db_add_mers(qo_db* db, const uint64_t* mers, const uint8_t* hq, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (qo_db_add(db, mers[i], hq[i])) return -1;
  return (int64_t)n;
}

#define QO_MAX_READ (1u << 24)

/* create_database main loop on one FASTQ buffer, single thread.
 * Returns the number of (mer, quality) instances added, or a negative parser
 * error (see qo_parse_fastq), or -10 for "Hash is full". */
int64_t qo_db_add_fastq(qo_db* db, const char* buf, size_t len, int qual_thresh) {
  size_t cap = len / 4 + 2;
  qo_rec* recs = (qo_rec*)malloc(cap * sizeof(qo_rec));
  char* scratch = (char*)malloc(len + 1);
  int64_t nrec = qo_parse_fastq(buf, len, recs, cap, scratch, len + 1);
  if (nrec < 0) { free(recs); free(scratch); return nrec; }
  uint32_t maxlen = 0;
  for (int64_t i = 0; i < nrec; ++i) if (recs[i].len > maxlen) maxlen = recs[i].len;
  uint64_t* mers = (uint64_t*)malloc((maxlen + 1) * sizeof(uint64_t));
  uint8_t* hq = (uint8_t*)malloc(maxlen + 1);
  uint64_t m = 0, rm = 0;
  int64_t total = 0;
  for (int64_t i = 0; i < nrec && total >= 0; ++i) {
    const char* base = recs[i].in_scratch ? scratch : buf;
    size_t n = qo_extract_read(base + recs[i].seq_off, base + recs[i].qual_off, recs[i].len,
                               db->k, qual_thresh, &m, &rm, mers, hq);
    for (size_t j = 0; j < n; ++j)
      if (qo_db_add(db, mers[j], hq[j])) { total = -10; break; }
    if (total >= 0) total += (int64_t)n;
  }
  free(mers); free(hq); free(recs); free(scratch);
  return total;
}

/* multithreaded form = the CPU baseline.  Threads pull jobs of 100 reads
 * (create_database.cc:52 "100 reads/job") off an atomic cursor and run the same
 * per-base loop + lock-free add.  The table must be sized up front (no growth
 * under concurrency here); returns -10 on "Hash is full". */
typedef struct {
  qo_db* db; const char* buf; const char* scratch; const qo_rec* recs; int64_t nrec;
  int thresh; int64_t* cursor; int64_t added; int failed;
} mt_arg;

static void* mt_worker(void* p) {
  mt_arg* a = (mt_arg*)p;
  uint64_t m = 0, rm = 0;
  size_t cap = 1024;
  uint64_t* mers = (uint64_t*)malloc(cap * sizeof(uint64_t));
  uint8_t* hq = (uint8_t*)malloc(cap);
  for (;;) {
    int64_t j0 = __atomic_fetch_add(a->cursor, 100, __ATOMIC_RELAXED);
    if (j0 >= a->nrec || a->failed) break;
    int64_t j1 = j0 + 100 < a->nrec ? j0 + 100 : a->nrec;
    for (int64_t i = j0; i < j1; ++i) {
      const qo_rec* r = &a->recs[i];
      if (r->len + 1 > cap) {
        cap = r->len + 1;
        mers = (uint64_t*)realloc(mers, cap * sizeof(uint64_t));
        hq = (uint8_t*)realloc(hq, cap);
      }
      const char* base = r->in_scratch ? a->scratch : a->buf;
      size_t n = qo_extract_read(base + r->seq_off, base + r->qual_off, r->len, a->db->k, a->thresh,
                                 &m, &rm, mers, hq);
      for (size_t t = 0; t < n; ++t)
        if (add_nogrow(a->db, mers[t], hq[t])) { a->failed = 1; break; }
      a->added += (int64_t)n;
    }
  }
  free(mers); free(hq);
  return 0;
}

int64_t qo_db_add_fastq_mt(qo_db* db, const char* buf, size_t len, int qual_thresh, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  size_t cap = len / 4 + 2;
  qo_rec* recs = (qo_rec*)malloc(cap * sizeof(qo_rec));
  char* scratch = (char*)malloc(len + 1);
  int64_t nrec = qo_parse_fastq(buf, len, recs, cap, scratch, len + 1);
  if (nrec < 0) { free(recs); free(scratch); return nrec; }
  int64_t cursor = 0;
  pthread_t th[256];
  mt_arg args[256];
  for (int t = 0; t < nthreads; ++t) {
    args[t] = (mt_arg){db, buf, scratch, recs, nrec, qual_thresh, &cursor, 0, 0};
    pthread_create(&th[t], 0, mt_worker, &args[t]);
  }
  int64_t total = 0; int failed = 0;
  for (int t = 0; t < nthreads; ++t) {
    pthread_join(th[t], 0);
    total += args[t].added; failed |= args[t].failed;
  }
  free(recs); free(scratch);
  return failed ? -10 : total;
}

/* ------------------------------------------------------------ database_query */

uint64_t qo_db_size(const qo_db* db) { return db->size; }
int qo_db_k(const qo_db* db) { return db->k; }
int qo_db_bits(const qo_db* db) { return db->bits; }

uint64_t qo_db_distinct(const qo_db* db) {
  uint64_t n = 0;
  for (uint64_t i = 0; i < db->size; ++i) n += db->keys[i] != 0;
  return n;
}

/* operator[] (mer_database.hpp:284-293) packed as count<<1 | hq; absent -> 0 */
uint64_t qo_db_lookup1(const qo_db* db, uint64_t mer) {
  int64_t id = key_get(db, mer);
  return id < 0 ? 0 : val_get(db, (uint64_t)id);
}

void qo_db_lookup(const qo_db* db, const uint64_t* mers, size_t n, uint64_t* out) {
  for (size_t i = 0; i < n; ++i) out[i] = qo_db_lookup1(db, mers[i]);
}

/* get_val (mer_database.hpp:296-299): hq ? count : 0 */
uint64_t qo_db_get_val(const qo_db* db, uint64_t mer) {
  uint64_t v = qo_db_lookup1(db, mer);
  return (v & 1) ? v >> 1 : 0;
}

/* get_best_alternatives (mer_database.hpp:302-329).  fmer is the forward mer of
 * a kmer_t; backward=0 means forward_mer (base 0 = the two low bits of fmer),
 * backward=1 means backward_mer (base 0 = base k-1 of fmer, kmer.hpp:101-103).
 * ucode is left untouched when nothing is kept, as upstream. */
int qo_db_best_alternatives(const qo_db* db, uint64_t fmer, int backward,
                            uint64_t counts[4], int* ucode, int* level) {
  const int k = db->k;
  const unsigned sh = backward ? 2 * (k - 1) : 0;
  int count = 0;
  memset(counts, 0, 4 * sizeof(uint64_t));
  *level = 0;
  for (int i = 0; i < 4; ++i) {
    uint64_t m = (fmer & ~(3ull << sh)) | ((uint64_t)i << sh);      /* m.replace(0, i) */
    uint64_t v = qo_db_lookup1(db, qo_canonical(m, k));
    uint64_t first = v >> 1; int second = (int)(v & 1);
    if (first > 0) {
      if (second >= *level) {
        if (second > *level && count > 0) {
          for (int j = 0; j < i; ++j) counts[j] = 0;
          count = 0;
        }
        counts[i] = first;
        *ucode = i;
        *level = second;
        ++count;
      }
    }
  }
  return count;
}

/* dump every (mer, count<<1|hq) in slot order; returns number written */
uint64_t qo_db_dump(const qo_db* db, uint64_t* mers, uint64_t* vals, uint64_t cap) {
  uint64_t n = 0;
  for (uint64_t i = 0; i < db->size && n < cap; ++i)
    if (db->keys[i]) { mers[n] = db->keys[i] - 1; vals[n] = val_get(db, i); ++n; }
  return n;
}

/* compute_poisson_cutoff__ reduction (error_correct_reads.cc:650-659) */
void qo_db_stats(const qo_db* db, uint64_t* distinct, uint64_t* total) {
  uint64_t d = 0, t = 0;
  for (uint64_t i = 0; i < db->size; ++i) {
    uint64_t v = val_get(db, i);
    if ((v & 1) && v >= 2) { d += 1; t += v >> 1; }
  }
  *distinct = d; *total = t;
}

/* histo_mer_database.cc:13-21: histos[min(count,1000)][hq] over occupied slots */
void qo_db_histo(const qo_db* db, uint64_t histo[1001][2]) {
  memset(histo, 0, 1001 * 2 * sizeof(uint64_t));
  for (uint64_t i = 0; i < db->size; ++i) {
    if (!db->keys[i]) continue;
    uint64_t v = val_get(db, i);
    uint64_t c = v >> 1; if (c > 1000) c = 1000;
    ++histo[c][v & 1];
  }
}

/* ---------------------------------------------------------- Poisson cutoff */

/* poisson_term (error_correct_reads.cc:53-61): e^-lambda lambda^i / i!, Stirling for i >= 11 */
double qo_poisson_term(double lambda, unsigned i) {
  static const double facts[11] = {1, 1, 2, 6, 24, 120, 720, 5040, 40320, 362880, 3628800};
  static const double tau = 6.283185307179583;
  if (i < 11) return exp(-lambda) * pow(lambda, i) / facts[i];
  return exp(-lambda + i) * pow(lambda / i, i) / sqrt(tau * i);
}

/* compute_poisson_cutoff__ (error_correct_reads.cc:650-668) from the (distinct, total) reduction */
unsigned qo_poisson_cutoff_from(uint64_t distinct, uint64_t total, double collision_prob, double poisson_threshold) {
  const double coverage = (double)total / (double)distinct;
  const double lambda = coverage * collision_prob;
  for (unsigned x = 2; x < 1000; ++x)
    if (qo_poisson_term(lambda, x) < poisson_threshold) return x + 1;
  return 0;
}

unsigned qo_db_poisson_cutoff(const qo_db* db, double collision_prob, double poisson_threshold) {
  uint64_t d, t;
  qo_db_stats(db, &d, &t);
  return qo_poisson_cutoff_from(d, t, collision_prob, poisson_threshold);
}

/* ------------------------------------------------- contaminant set (FASTA) */

/* The set `jellyfish count -C -m k` makes of a FASTA file (Makefile.am:54-55) and
 * contaminant_database loads key by key (error_correct_reads.cc:86-90): every canonical
 * k-mer of every record, a non-ACGT character restarting the window.  FASTA as Jellyfish's
 * whole_sequence_parser::read_fasta reads it: '>' header line, then sequence lines up to the
 * next line that starts with '>'.  Members are stored with quality 1, so membership
 * (mer_array::has_key, :94) is "lookup != 0".  Returns the number of instances, -4 when the
 * first character is not '>', -10 when the table is full. */
int64_t qo_db_add_fasta(qo_db* db, const char* buf, size_t len) {
  if (len == 0) return 0;
  if (buf[0] != '>') return -4;
  char* seq = (char*)malloc(len + 1);
  char* qual = (char*)malloc(len + 1);
  uint64_t* mers = (uint64_t*)malloc((len + 1) * sizeof(uint64_t));
  uint8_t* hq = (uint8_t*)malloc(len + 1);
  memset(qual, 127, len + 1);
  size_t p = 0;
  int64_t total = 0;
  uint64_t m = 0, rm = 0;
  while (p < len && total >= 0) {
    p = line_end(buf, len, p);            /* header line */
    if (p < len) ++p;
    size_t n = 0;
    while (p < len && buf[p] != '>') {    /* sequence lines */
      size_t e = line_end(buf, len, p);
      memcpy(seq + n, buf + p, e - p);
      n += e - p;
      p = e < len ? e + 1 : e;
    }
    size_t cnt = qo_extract_read(seq, qual, n, db->k, 0, &m, &rm, mers, hq);
    for (size_t j = 0; j < cnt; ++j)
      if (qo_db_add(db, mers[j], 1)) { total = -10; break; }
    if (total >= 0) total += (int64_t)cnt;
  }
  free(seq); free(qual); free(mers); free(hq);
  return total;
}
