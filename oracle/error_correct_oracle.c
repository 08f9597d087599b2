/*
 * error_correct_oracle.c -- CPU restatement of Quorum's read corrector
 * (quorum_error_correct_reads), the first "next" row of SURVEY.md 8(f).
 *
 * TEST INFRASTRUCTURE ONLY (same rules as quorum_oracle.c): used by tests/ and by
 * the cpu_baseline leg of tools/correct_bench.py as the checker / reported CPU
 * baseline.  Nothing in the product path may include, link or call it.
 *
 * What it restates (paths relative to the upstream Quorum tree):
 *   - error_correct_instance::start      src/error_correct_reads.cc:246-341
 *   - find_starting_mer                  src/error_correct_reads.cc:609-643
 *   - extend (both directions)           src/error_correct_reads.cc:386-565
 *   - log_substitution, check_contaminant  :347-379
 *   - homo_trim                          src/error_correct_reads.cc:567-597
 *   - err_log<T> (window rule, remove_last_window, force_truncate, printing)
 *                                        src/err_log.hpp:55-135
 *   - forward/backward counters, pointers, logs   src/error_correct_reads.hpp:15-173
 *   - kmer_t, forward_mer, backward_mer  src/kmer.hpp:11-106
 *
 * The two directions are one function here: a direction d = +1 / -1 and the
 * "directional coordinate" d * raw turn every forward_counter / backward_counter
 * operator into ordinary integer arithmetic:
 *     ++pos  -> raw += d          pos + x -> raw + d*x      a - b -> d*(a.raw - b.raw)
 *     a > b  -> d*a.raw > d*b.raw   and an `unsigned` compared with a counter is first
 *     converted to a counter (implicit constructor), so  pos > _window  is
 *     d*raw > d*window  (for the backward log: raw < window, err_log.hpp:89).
 *
 * PARITY UNPINNED: the reference holds no test, fixture or golden output for the
 * corrector (unit_tests/ only covers mer_database).  The only reference-held example
 * is the output format in README.md:58-88, which tests/test_correct_oracle.py checks
 * the printer against.  Everything else is pinned by hand-derived cases worked from
 * the sources above.  Known deliberate deviation: kmer_t::shift_left(char) tests
 * `!= -1` (kmer.hpp:15-22), so in find_starting_mer a character whose Jellyfish code is
 * -2/-3 (neither a base nor N/IUPAC) is shifted in as a negative code upstream; here
 * every non-ACGT character restarts the anchor search the way N does.
 */
#define _GNU_SOURCE
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct qo_db qo_db;
extern uint64_t qo_db_lookup1(const qo_db* db, uint64_t mer);
extern uint64_t qo_db_get_val(const qo_db* db, uint64_t mer);
extern int qo_db_best_alternatives(const qo_db* db, uint64_t fmer, int backward, uint64_t counts[4], int* ucode,
                                   int* level);
extern int qo_db_k(const qo_db* db);
extern uint64_t qo_canonical(uint64_t m, int k);
extern int qo_code(int c);
extern double qo_poisson_term(double lambda, unsigned i);

typedef struct qo_ec_params {
  int skip, good, anchor, min_count, cutoff; /* -s -g -a -m -p (error_correct_reads_cmdline.yaggo:9-20,51-53) */
  int window, error;                         /* -w -e; 0 means k and k/2 (error_correct_reads.cc:206-207) */
  int qual_cutoff;                           /* compared as char; CHAR_MAX when not given (:684-685) */
  int trim_contaminant;                      /* --trim-contaminant */
  int do_homo_trim, homo_trim;               /* --homo-trim given?, its value */
  int no_discard;                            /* -d */
  double collision_prob;                     /* apriori_error_rate / 3 (:734) */
  double poisson_threshold;                  /* :735 */
} qo_ec_params;

/* --------------------------------------------------------------- kmer_t */

typedef struct { uint64_t f, r; } kmer;

static inline uint64_t kmask(int k) { return (1ull << (2 * k)) - 1; }
static void shift_left(kmer* m, int k, int x) { /* kmer.hpp:24-27 */
  m->f = ((m->f << 2) | (uint64_t)x) & kmask(k);
  m->r = (m->r >> 2) | ((uint64_t)(3 - x) << (2 * (k - 1)));
}
static void shift_right(kmer* m, int k, int x) { /* kmer.hpp:38-41 */
  m->f = (m->f >> 2) | ((uint64_t)x << (2 * (k - 1)));
  m->r = ((m->r << 2) | (uint64_t)(3 - x)) & kmask(k);
}
static void kreplace(kmer* m, int k, int i, int x) { /* kmer.hpp:47-50 */
  m->f = (m->f & ~(3ull << (2 * i))) | ((uint64_t)x << (2 * i));
  const int j = k - i - 1;
  m->r = (m->r & ~(3ull << (2 * j))) | ((uint64_t)(3 - x) << (2 * j));
}
static uint64_t kcanonical(const kmer* m) { return m->f < m->r ? m->f : m->r; } /* kmer.hpp:43 */
/* forward_mer / backward_mer (kmer.hpp:70-106): d = +1 forward, -1 backward */
static void dshift(kmer* m, int k, int d, int x) { if (d > 0) shift_left(m, k, x); else shift_right(m, k, x); }
static void dreplace0(kmer* m, int k, int d, int x) { kreplace(m, k, d > 0 ? 0 : k - 1, x); }
static int dcode0(const kmer* m, int k, int d) { return (int)((m->f >> (d > 0 ? 0 : 2 * (k - 1))) & 3); }

/* --------------------------------------------------------------- err_log */

enum { SUBSTITUTION = 0, TRUNCATION = 1 };
typedef struct { int type, pos; char from, to; } ec_entry;
typedef struct {
  int d;                  /* +1: err_log<forward_counter> "3_trunc", -1: err_log<backward_counter> "5_trunc" */
  unsigned window, error;
  size_t lwin, n, cap;
  ec_entry* e;
} ec_log;

static void log_init(ec_log* l, int d, unsigned window, unsigned error, ec_entry* store, size_t cap) {
  l->d = d; l->window = window; l->error = error; l->lwin = 0; l->n = 0; l->cap = cap; l->e = store;
}

/* err_log.hpp:87-95 */
static int log_check_nb_error(ec_log* l) {
  const int d = l->d;
  if (l->n > 0 && d * l->e[l->n - 1].pos > d * (int)l->window)
    while (d * l->e[l->n - 1].pos > d * l->e[l->lwin].pos + (int)l->window) l->lwin++;
  return (size_t)(l->n - l->lwin - 1) >= (size_t)l->error;
}
static int log_push(ec_log* l, int type, int pos, char from, char to) {
  if (l->n >= l->cap) abort(); /* capacity is len + 4: cannot happen */
  ec_entry* e = &l->e[l->n++];
  e->type = type; e->pos = pos; e->from = from; e->to = to;
  return log_check_nb_error(l);
}
static int log_substitution_entry(ec_log* l, int pos, char from, char to) { return log_push(l, SUBSTITUTION, pos, from, to); }
/* err_log::truncation (err_log.hpp:68-72); backward_log::truncation logs pos - 1 in
 * backward-counter arithmetic = raw + 1 (error_correct_reads.hpp:170-172) */
static int log_truncation_base(ec_log* l, int pos) { return log_push(l, TRUNCATION, pos, 0, 0); }
static int log_truncation(ec_log* l, int pos) { return log_truncation_base(l, l->d > 0 ? pos : pos + 1); }
/* err_log.hpp:97-106 */
static int log_remove_last_window(ec_log* l) {
  if (l->n == 0) return 0;
  const int diff = l->d * (l->e[l->n - 1].pos - l->e[l->lwin].pos);
  l->n = l->lwin;
  l->lwin = 0;
  log_check_nb_error(l);
  return diff;
}
/* err_log.hpp:75-83: drops every entry with  e.pos >= T(pos)  in the counter's own order */
static void log_force_truncate(ec_log* l, int pos) {
  size_t w = 0;
  for (size_t i = 0; i < l->n; ++i)
    if (!(l->d * l->e[i].pos >= l->d * pos)) l->e[w++] = l->e[i];
  l->n = w;
  l->lwin = 0;
  log_check_nb_error(l);
}
/* operator<< (err_log.hpp:111-135) */
static size_t log_print(const ec_log* l, char* out) {
  size_t n = 0;
  for (size_t i = 0; i < l->n; ++i) {
    if (i) out[n++] = ' ';
    if (l->e[i].type == SUBSTITUTION)
      n += (size_t)sprintf(out + n, "%d:sub:%c-%c", l->e[i].pos, l->e[i].from, l->e[i].to);
    else
      n += (size_t)sprintf(out + n, "%d:%s", l->e[i].pos, l->d > 0 ? "3_trunc" : "5_trunc");
  }
  return n;
}

/* --------------------------------------------------------------- corrector */

enum { EC_OK = 0, EC_TRUNCATE = 1, EC_ERROR = 2 };
static const char* const error_contaminant = "Contaminated read";
static const char* const error_no_starting_mer = "No high quality mer";
static const char* const error_homopolymer = "Entire read is an homopolymer";

typedef struct {
  const qo_db* db;
  const qo_db* contaminant; /* NULL = contaminant_no_database */
  const qo_ec_params* p;
  int k, window, error;
} ec_ctx;

static int is_contaminant(const ec_ctx* c, uint64_t canon) {
  return c->contaminant && qo_db_lookup1(c->contaminant, canon) != 0;
}

/* error_correct_reads.cc:346-357 */
static int check_contaminant(const ec_ctx* c, const kmer* mer, ec_log* log, int cpos, const char** error) {
  if (is_contaminant(c, kcanonical(mer))) {
    if (c->p->trim_contaminant) { log_truncation(log, cpos); return EC_TRUNCATE; }
    *error = error_contaminant;
    return EC_ERROR;
  }
  return EC_OK;
}

/* error_correct_reads.cc:360-379; *out is the (directional) output index */
static int log_substitution(const ec_ctx* c, kmer* mer, int d, int* out, ec_log* log, int cpos, int from, int to,
                            const char** error) {
  if (from == to) return EC_OK;
  dreplace0(mer, c->k, d, to);
  const int cc = check_contaminant(c, mer, log, cpos, error);
  if (cc != EC_OK) return cc;
  if (log_substitution_entry(log, cpos, from >= 0 ? "ACGT"[from] : 'N', to >= 0 ? "ACGT"[to] : 'N')) {
    const int diff = log_remove_last_window(log);
    *out -= d * diff;
    log_truncation(log, cpos - d * diff);
    return EC_TRUNCATE;
  }
  return EC_OK;
}

/* error_correct_reads.cc:386-565.  in/end/out are indices into seq / the output buffer;
 * the quality pointer moves in lockstep with the input pointer, so *qual is qual[in].
 * Returns the output index (out.ptr()) or INT_MIN for "return 0". */
static int extend(const ec_ctx* c, kmer mer, int d, const char* seq, const char* qual, int in, int pos, int end,
                  int out, char* buffer, ec_log* log, const char** error) {
  const int k = c->k;
  const qo_ec_params* p = c->p;
  int cpos = pos;
  uint32_t prev_count = (uint32_t)qo_db_get_val(c->db, kcanonical(&mer));

  for (; d * in < d * end; in += d) {
    const char base = seq[in];
    if (base == '\n') continue;
    cpos = pos;
    pos += d;

    const int ori_code = qo_code((unsigned char)base);
    dshift(&mer, k, d, ori_code >= 0 ? ori_code : 0);
    if (ori_code >= 0) {
      const int cc = check_contaminant(c, &mer, log, cpos, error);
      if (cc == EC_TRUNCATE) goto done;
      if (cc == EC_ERROR) return INT_MIN;
    }
    uint64_t counts[4];
    int ucode = 0, level;
    const int count = qo_db_best_alternatives(c->db, mer.f, d < 0, counts, &ucode, &level);

    if (count == 0) { log_truncation(log, cpos); goto done; }

    if (count == 1) {
      prev_count = (uint32_t)counts[ucode];
      const int lc = log_substitution(c, &mer, d, &out, log, cpos, ori_code, ucode, error);
      if (lc == EC_TRUNCATE) goto done;
      if (lc == EC_ERROR) return INT_MIN;
      buffer[out] = "ACGT"[dcode0(&mer, k, d)];
      out += d;
      continue;
    }
    if (ori_code >= 0) {
      if (counts[ori_code] > (uint64_t)p->min_count) {
        if (counts[ori_code] >= (uint32_t)p->cutoff || (char)qual[in] >= (char)p->qual_cutoff) {
          buffer[out] = "ACGT"[dcode0(&mer, k, d)];
          out += d;
          continue;
        }
        const double pp = (double)(counts[0] + counts[1] + counts[2] + counts[3]) * p->collision_prob;
        const double prob = qo_poisson_term(pp, (unsigned)counts[ori_code]);
        if (prob < p->poisson_threshold) {
          buffer[out] = "ACGT"[dcode0(&mer, k, d)];
          out += d;
          continue;
        }
      } else if (level == 0 && counts[ori_code] == 0) {
        log_truncation(log, cpos);
        goto done;
      }
    } else if (level == 0) {
      log_truncation(log, cpos);
      goto done;
    }

    int check_code = ori_code;
    int success = 0;
    uint64_t cont_counts[4];
    int continue_with_correct_base[4];
    int read_nbase_code = -1;
    int candidate_continuations[4];
    unsigned ncandidate_continuations = 0;

    if (d * (in + d) < d * end) read_nbase_code = qo_code((unsigned char)seq[in + d]);

    for (int i = 0; i < 4; ++i) {
      cont_counts[i] = 0;
      continue_with_correct_base[i] = 0;
      if (counts[i] <= (uint64_t)p->min_count) continue;
      check_code = i;
      kmer nmer = mer;
      dreplace0(&nmer, k, d, check_code);
      dshift(&nmer, k, d, 0);
      uint64_t ncounts[4];
      int nucode = 0, nlevel;
      const int ncount = qo_db_best_alternatives(c->db, nmer.f, d < 0, ncounts, &nucode, &nlevel);
      if (ncount > 0 && nlevel >= level) {
        continue_with_correct_base[i] = read_nbase_code >= 0 && ncounts[read_nbase_code] > 0;
        success = 1;
        cont_counts[i] = counts[i];
      }
    }

    if (success) {
      check_code = -1;
      const uint32_t _prev_count = prev_count <= (uint64_t)p->min_count ? UINT32_MAX : prev_count;
      int min_diff = INT_MAX;
      for (int i = 0; i < 4; ++i) {
        candidate_continuations[i] = 0;
        if (cont_counts[i] > 0) {
          const int dd = (int)labs((long)cont_counts[i] - (long)_prev_count); /* the (int) cast is upstream's, :521 */
          if (dd < min_diff) min_diff = dd;
        }
      }
      for (uint32_t i = 0; i < 4; i++) {
        if (labs((long)cont_counts[i] - (long)_prev_count) == (long)min_diff) { /* long == int, :526 */
          candidate_continuations[i] = 1;
          ++ncandidate_continuations;
          check_code = (int)i;
        }
      }
      if (ncandidate_continuations > 1 && read_nbase_code >= 0)
        for (int i = 0; i < 4; ++i) {
          if (candidate_continuations[i]) {
            if (!continue_with_correct_base[i]) --ncandidate_continuations;
            else check_code = i;
          }
        }
      if (ncandidate_continuations != 1) check_code = -1;
      if (check_code >= 0) {
        const int lc = log_substitution(c, &mer, d, &out, log, cpos, ori_code, check_code, error);
        if (lc == EC_TRUNCATE) goto done;
        if (lc == EC_ERROR) return INT_MIN;
      }
    }
    if (ori_code < 0 && check_code < 0) {
      log_truncation(log, cpos);
      goto done;
    }
    buffer[out] = "ACGT"[dcode0(&mer, k, d)];
    out += d;
  }
done:
  return out;
}

/* error_correct_reads.cc:609-643.  *in / *out advance together (they are equal as indices). */
static int find_starting_mer(const ec_ctx* c, kmer* mer, const char* seq, int* in, int end, char* buffer,
                             const char** error) {
  const int k = c->k;
  while (*in < end) {
    for (int i = 0; *in < end && i < k; ++i) {
      const char base = seq[(*in)];
      buffer[(*in)++] = base;
      const int x = qo_code((unsigned char)base);
      if (x >= 0) shift_left(mer, k, x);
      else i = -1;
    }
    int found = 0;
    while (*in < end) {
      const int contaminated = is_contaminant(c, kcanonical(mer));
      if (contaminated && !c->p->trim_contaminant) { *error = error_contaminant; return 0; }
      if (!contaminated) {
        const uint64_t val = qo_db_get_val(c->db, kcanonical(mer));
        found = (int)val >= c->p->anchor ? found + 1 : 0;
        if (found >= c->p->good) return 1;
      }
      const char base = seq[(*in)];
      buffer[(*in)++] = base;
      const int x = qo_code((unsigned char)base);
      if (x < 0) break;
      shift_left(mer, k, x);
    }
  }
  *error = error_no_starting_mer;
  return 0;
}

/* error_correct_reads.cc:567-597; indices into buffer. Returns new end, or INT_MIN. */
static int homo_trim(const ec_ctx* c, const char* buffer, int out_start, int out_end, ec_log* fwd, ec_log* bwd,
                     const char** error) {
  int max_homo_score = INT_MIN, max_pos = INT_MIN, homo_score = 0;
  int ptr = out_end - 1;
  char pbase = (char)qo_code((unsigned char)buffer[ptr]);
  for (--ptr; ptr >= out_start; --ptr) {
    const char cbase = (char)qo_code((unsigned char)buffer[ptr]);
    homo_score += ((pbase == cbase) << 1) - 1;
    pbase = cbase;
    if (homo_score > max_homo_score) { max_homo_score = homo_score; max_pos = ptr; }
  }
  if (max_homo_score < c->p->homo_trim) return out_end;
  if (max_pos < out_start) { *error = error_homopolymer; return INT_MIN; }
  log_force_truncate(fwd, max_pos);
  log_force_truncate(bwd, max_pos);
  log_truncation_base(fwd, max_pos); /* fwd_log.truncation(forward_counter(..)) */
  return max_pos;
}

typedef struct { char* p; size_t n, cap; int overflow; } sbuf;
static void sb_put(sbuf* b, const char* s, size_t n) {
  if (b->n + n > b->cap) { b->overflow = 1; return; }
  memcpy(b->p + b->n, s, n);
  b->n += n;
}
static void sb_puts(sbuf* b, const char* s) { sb_put(b, s, strlen(s)); }

/* The body of the per-read loop of error_correct_instance::start (:255-337). */
static void correct_read(const ec_ctx* c, const char* header, size_t hlen, const char* seq, const char* qual, int len,
                         char* buffer, ec_entry* store, char* tmp, sbuf* fa, sbuf* details) {
  const int k = c->k;
  const char* error = "";
  int in = c->p->skip;
  kmer mer = {0, 0};
  ec_log fwd_log, bwd_log;
#define SKIP_READ()                                              \
  do {                                                           \
    sb_puts(details, "Skipped "); sb_put(details, header, hlen); \
    sb_puts(details, ": "); sb_puts(details, error); sb_puts(details, "\n"); \
    if (c->p->no_discard) { sb_puts(fa, ">"); sb_put(fa, header, hlen); sb_puts(fa, "\nN\n"); } \
    return;                                                      \
  } while (0)

  if (!find_starting_mer(c, &mer, seq, &in, len, buffer, &error)) SKIP_READ();
  const int start_off = in;
  log_init(&fwd_log, +1, (unsigned)c->window, (unsigned)c->error, store, (size_t)len + 4);
  int end_out = extend(c, mer, +1, seq, qual, in, start_off, len, in, buffer, &fwd_log, &error);
  if (end_out == INT_MIN) SKIP_READ();
  log_init(&bwd_log, -1, (unsigned)c->window, (unsigned)c->error, store + len + 4, (size_t)len + 4);
  int start_out = extend(c, mer, -1, seq, qual, in - k - 1, start_off - k - 1, -1, in - k - 1, buffer, &bwd_log, &error);
  if (start_out == INT_MIN) SKIP_READ();
  start_out++;
  if (c->p->do_homo_trim) {
    end_out = homo_trim(c, buffer, start_out, end_out, &fwd_log, &bwd_log, &error);
    if (end_out == INT_MIN) SKIP_READ();
  }
  sb_puts(fa, ">"); sb_put(fa, header, hlen); sb_puts(fa, " ");
  sb_put(fa, tmp, log_print(&fwd_log, tmp));
  sb_puts(fa, " ");
  sb_put(fa, tmp, log_print(&bwd_log, tmp));
  sb_puts(fa, "\n");
  if (end_out > start_out) sb_put(fa, buffer + start_out, (size_t)(end_out - start_out));
  sb_puts(fa, "\n");
#undef SKIP_READ
}

/* FASTQ records with their headers: same rules as qo_parse_fastq (quorum_oracle.c). */
typedef struct { uint64_t seq_off, qual_off; uint32_t len; uint32_t in_scratch; } qo_rec;
extern int64_t qo_parse_fastq_hdr(const char* buf, size_t len, qo_rec* recs, size_t cap, char* scratch,
                                  size_t scratch_cap, uint64_t* hdr_off, uint32_t* hdr_len);

/* quorum_error_correct_reads over one FASTQ buffer with one thread: the corrected
 * reads (".fa" stream) and the skip log (".log" stream).  Returns the number of reads,
 * or a negative qo_parse_fastq error, or -5 when an output buffer is too small. */
int64_t qo_ec_correct_fastq(const qo_db* db, const qo_db* contaminant, const qo_ec_params* p, const char* buf, size_t len,
                            char* fa_out, size_t fa_cap, size_t* fa_len, char* log_out, size_t log_cap, size_t* log_len) {
  ec_ctx c;
  c.db = db; c.contaminant = contaminant; c.p = p; c.k = qo_db_k(db);
  c.window = p->window ? p->window : c.k;      /* :206 */
  c.error = p->error ? p->error : c.k / 2;     /* :207 */
  const size_t cap = len / 4 + 2;
  qo_rec* recs = (qo_rec*)malloc(cap * sizeof(qo_rec));
  uint64_t* hoff = (uint64_t*)malloc(cap * sizeof(uint64_t));
  uint32_t* hlen = (uint32_t*)malloc(cap * sizeof(uint32_t));
  char* scratch = (char*)malloc(len + 1);
  const int64_t n = qo_parse_fastq_hdr(buf, len, recs, cap, scratch, len + 1, hoff, hlen);
  sbuf fa = {fa_out, 0, fa_cap, 0}, details = {log_out, 0, log_cap, 0};
  if (n > 0) {
    size_t maxlen = 0;
    for (int64_t i = 0; i < n; ++i) if (recs[i].len > maxlen) maxlen = recs[i].len;
    char* buffer = (char*)malloc(maxlen + 16);
    ec_entry* store = (ec_entry*)malloc((2 * maxlen + 8) * sizeof(ec_entry));
    char* tmp = (char*)malloc((maxlen + 4) * 24 + 16);
    for (int64_t i = 0; i < n; ++i) {
      const char* base = recs[i].in_scratch ? scratch : buf;
      /* store is laid out for this read's length (two logs of len + 4 entries) */
      correct_read(&c, buf + hoff[i], hlen[i], base + recs[i].seq_off, base + recs[i].qual_off, (int)recs[i].len,
                   buffer, store, tmp, &fa, &details);
    }
    free(buffer); free(store); free(tmp);
  }
  free(recs); free(hoff); free(hlen); free(scratch);
  *fa_len = fa.n; *log_len = details.n;
  if (n >= 0 && (fa.overflow || details.overflow)) return -5;
  return n;
}

/* the log printer alone, for the README.md:58-88 format check:
 * entries = {type, pos, from, to} quadruples */
size_t qo_ec_print_log(int d, const int* entries, size_t n, char* out) {
  ec_log l;
  ec_entry* e = (ec_entry*)malloc((n + 1) * sizeof(ec_entry));
  log_init(&l, d, 1, 1, e, n);
  for (size_t i = 0; i < n; ++i) {
    e[i].type = entries[4 * i]; e[i].pos = entries[4 * i + 1];
    e[i].from = (char)entries[4 * i + 2]; e[i].to = (char)entries[4 * i + 3];
  }
  l.n = n;
  const size_t r = log_print(&l, out);
  free(e);
  return r;
}
