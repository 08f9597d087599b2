// qh_correct.hip -- Quorum's read corrector (quorum_error_correct_reads) on gfx950: SURVEY 8(f)-1.
//
// Replaces (paths relative to the Quorum tree):
//   error_correct_instance::start       src/error_correct_reads.cc:246-341
//   find_starting_mer                   src/error_correct_reads.cc:609-643
//   extend, both directions             src/error_correct_reads.cc:386-565
//   log_substitution, check_contaminant src/error_correct_reads.cc:346-379
//   homo_trim                           src/error_correct_reads.cc:567-597
//   err_log<T> + its printer            src/err_log.hpp:55-135
//   forward/backward counters and logs  src/error_correct_reads.hpp:87-173
//   kmer_t, forward_mer, backward_mer   src/kmer.hpp:11-106
//   the two output streams              src/error_correct_reads.cc:272-275,334-336
//
// One read per lane. A read is a serial chain of dependent table lookups (4 per base, 16 more
// where several alternatives survive), so the kernel is bound by the random 8-byte load rate
// of HBM (measured 54 G/s, qh_bench_random_access mode 1) and wants as many reads in flight as
// registers allow; nothing here is MFMA-shaped. Both directions of extend() run in ONE loop
// with the direction as lane state (d = +1 / -1), so a lane that finishes its forward
// extension early goes on backwards while its neighbours are still extending forwards.
// The two counters of error_correct_reads.hpp become "directional coordinates" d * raw: every
// forward_counter / backward_counter operator is then plain integer arithmetic (an `unsigned`
// compared with a counter is converted to a counter first, so `pos > _window` of
// err_log.hpp:89 is d*raw > d*window).
// The error logs live in global memory (at most len + 4 entries per read, entry = one word).
// The text of both output streams is produced on the device too (sizes -> scan -> write), in
// input order, so the host only copies two byte strings out.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <string.h>
#include <string>
#include <vector>

#include "qh_internal.h"
#include "qh_device.h"

#define EC_BLOCK 256
#ifndef EC_WAVES
#define EC_WAVES 4  // waves per SIMD the corrector kernel is compiled for: 128 VGPRs, a few spilled words (measured: 3 -> 4 waves 8.9 -> 7.5 ms)
#endif
#define EC_FMT_BLOCK 256
#define EC_RES_WORDS 8  // per read: status, start, end, nf, nb, offset of the backward log, fa_bytes, log_bytes

enum { EC_ST_OK = 0, EC_ST_CONTAMINANT = 1, EC_ST_NO_START = 2, EC_ST_HOMOPOLYMER = 3 };
enum { EC_OK = 0, EC_TRUNCATE = 1, EC_ERROR = 2 };

struct qh_ec_dev {
  qh_geom g;
  const uint64_t* table;
  qh_geom cg;
  const uint64_t* ctable;  // null: contaminant_no_database
  // bit [h & cbits_mask] is set for the hash h of every contaminant k-mer (cbits_mask == 0: no such filter). A
  // contaminant set is a few thousand k-mers and nearly every base of every read is not one of them: the
  // bit answers that without the walk through the set's table (a 64-byte line and eight tag compares per base).
  const uint32_t* cbits;
  uint32_t cbits_mask;
  int skip, good, anchor, min_count, cutoff, window, error, qual_cutoff;
  int trim_contaminant, do_homo_trim, homo_trim, no_discard;
  // bit [c * ptab_stride + sum]: poisson_term(sum * collision_prob, c) < poisson_threshold, evaluated by the
  // host's libm (error_correct_reads.cc:448-450); ptab_stride == 0: evaluate on the device
  int ptab_stride;
  const uint32_t* ptab;
  double collision_prob, poisson_threshold;
};

struct qh_ec_totals {
  unsigned long long fa_bytes, log_bytes;
  unsigned long long fa_cap, log_cap;  // capacity the format kernel may write
  unsigned long long reads, corrected, skipped, substitutions, truncations, bases_in, bases_out;
  unsigned int overflow;               // a stream did not fit its buffer: nothing of it was written
};

// ------------------------------------------------------------------ lookups

// The k-mer under the read head as bit planes, the form the table hashes (qh_spec.h: plane key =
// low code bits | high code bits << k): a0/a1 = low/high code bits of the strand that is being
// extended at its low end, b0/b1 of its reverse complement, base i at bit i. Extending backwards on
// the forward strand IS extending forwards on the reverse strand with complemented codes, so one
// direction-free set of operations serves forward_mer and backward_mer (kmer.hpp:70-106): a = the
// forward mer and y = the base code when d > 0, a = the reverse complement and y = 3 - code when d < 0.
// The 2-bit interleaved mer is never built: shifting a base in costs 8 32-bit operations, the
// canonical form ~10 (qh_canonical_planes) and the position hash is the table's own 3 multiplies.
struct ec_mer { uint32_t a0, a1, b0, b1; };

__device__ __forceinline__ void ec_shift(ec_mer& m, int k, uint32_t km, uint32_t y) {
  const uint32_t ny = y ^ 3u;
  m.a0 = ((m.a0 << 1) | (y & 1u)) & km;
  m.a1 = ((m.a1 << 1) | (y >> 1)) & km;
  m.b0 = (m.b0 >> 1) | ((ny & 1u) << (k - 1));
  m.b1 = (m.b1 >> 1) | ((ny >> 1) << (k - 1));
}
__device__ __forceinline__ void ec_replace0(ec_mer& m, int k, uint32_t y) {  // kmer.hpp:47-50
  const uint32_t ny = y ^ 3u, top = 1u << (k - 1);
  m.a0 = (m.a0 & ~1u) | (y & 1u);
  m.a1 = (m.a1 & ~1u) | (y >> 1);
  m.b0 = (m.b0 & ~top) | ((ny & 1u) << (k - 1));
  m.b1 = (m.b1 & ~top) | ((ny >> 1) << (k - 1));
}
// hash of mer.canonical() (kmer.hpp:43): position and remainder of the key in any table of this k
__device__ __forceinline__ uint64_t ec_hash(const ec_mer& m, int k) {
  return qh_hash_strands(m.a0, m.a1, m.b0, m.b1, k);
}
__device__ __forceinline__ uint32_t ec_code0(const ec_mer& m) { return (m.a0 & 1u) | ((m.a1 & 1u) << 1); }

// WM, the word mode of a kernel: 0 = the value field and the remainder + probe fields (the tag) each fit 32 bits
// (k = 24 at genome-size tables): counts and tag compares are 32-bit; 1 = 64-bit tags, 32-bit values (k = 31, and
// every run with a contaminant set, whose small table has long tags): a slot is 64 bits, so a tag of more than 32
// bits leaves a value of fewer; 2 = everything 64-bit (a value field of more than 32 bits: -b above 31); 3 = like 2,
// and a table may be in the wide form (qh_spec.h): a slot that matches names the word of the side array that holds
// the value (the compatibility route for geometries whose key and value do not share one word).
template <int WM> struct ec_tagw { typedef uint64_t type; };
template <> struct ec_tagw<0> { typedef uint32_t type; };
template <int WM> struct ec_valw { typedef uint32_t type; };
template <> struct ec_valw<2> { typedef uint64_t type; };
template <> struct ec_valw<3> { typedef uint64_t type; };

// database_query::operator[] for N hashed mers at once. Probing is linear inside a 2^wbits-slot
// window, so consecutive probes are consecutive words: every round loads the 64-byte line of eight
// slots that holds the next probes of each unresolved lookup, and all N lookups advance in the same
// round. What bounds a read is the number of dependent memory round trips per base; three of the four
// alternatives of get_best_alternatives are normally ABSENT, and an unsuccessful search at load 0.55
// walks 2.9 slots on average with a long tail, of which the wave pays the maximum over its 64 lanes
// x 4 lookups. One slot per round and one lookup after the other cost about 60 round trips per base;
// four slots per round 6.6 (simulated maximum over a wave), eight slots per round 3.9. A line is the
// unit HBM delivers anyway, so the wider rounds add no memory traffic.
#define EC_GROUP 8
struct ec_group { uint64_t s[EC_GROUP]; };

template <int N, int WM>
__device__ __forceinline__ void ec_lookup_wide(const qh_geom& g, const uint64_t* __restrict__ table,
                                               const uint64_t (&h)[N], typename ec_valw<WM>::type (&v)[N]) {
  typedef typename ec_tagw<WM>::type W;   // tags
  typedef typename ec_valw<WM>::type V;   // values
  const uint32_t wmask = (1u << g.wbits) - 1u;
  uint64_t gaddr[N];  // slot index of the group's first slot
  W tag0[N];
  int pi[N];          // probe index of the group's first slot (negative: slots before the home slot)
  bool pend[N];
  ec_group grp[N];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    const qh_pos p = qh_locate_hash(g, h[n]);
    v[n] = 0;
    pend[n] = p.shard == (uint32_t)g.shard_id;
    const uint32_t ho = (uint32_t)p.home & wmask;
    tag0[n] = (W)p.tag0;
    gaddr[n] = p.home & ~(uint64_t)(EC_GROUP - 1);
    pi[n] = -(int)(ho & (uint32_t)(EC_GROUP - 1));
    if (pend[n]) grp[n] = *reinterpret_cast<const ec_group*>(table + gaddr[n]);
  }
  for (;;) {
    bool any = false;
#pragma unroll
    for (int n = 0; n < N; ++n) {
      if (!pend[n]) continue;
#pragma unroll
      for (int j = 0; j < EC_GROUP; ++j) {
        const int idx = pi[n] + j;
        const uint64_t sj = grp[n].s[j];
        if (pend[n] && idx >= 0) {
          if (idx > g.max_reprobe || sj == 0) pend[n] = false;
          else if ((W)(sj >> g.vbits) == (tag0[n] | (W)(idx + 1))) {
            v[n] = (WM == 3 && g.wide) ? (V)g.wvals[gaddr[n] + (uint64_t)j] : (V)sj & (V)g.vmask;
            pend[n] = false;
          }
        }
      }
      if (pend[n]) {  // next group of the window (wraps at the window's end)
        gaddr[n] = (gaddr[n] & ~(uint64_t)wmask) | (((uint32_t)gaddr[n] + (uint32_t)EC_GROUP) & wmask);
        pi[n] += EC_GROUP;
        any = true;
      }
    }
    if (!any) break;
#pragma unroll
    for (int n = 0; n < N; ++n)
      if (pend[n]) grp[n] = *reinterpret_cast<const ec_group*>(table + gaddr[n]);
  }
}

// tables of fewer than EC_GROUP slots per window: one slot at a time
__device__ __forceinline__ uint64_t ec_lookup_narrow_table(const qh_geom& g, const uint64_t* __restrict__ table, uint64_t h) {
  const qh_pos p = qh_locate_hash(g, h);
  if (p.shard != (uint32_t)g.shard_id) return 0;
  const uint64_t wmask = qh_low_mask(g.wbits);
  const uint64_t base = p.home & ~wmask;
  for (uint32_t i = 0; i <= (uint32_t)g.max_reprobe; ++i) {
    const uint64_t at = base | ((p.home + i) & wmask);
    const uint64_t s = table[at];
    if (s == 0) return 0;
    if ((s >> g.vbits) == (p.tag0 | (i + 1))) return qh_slot_val(g, s, at);
  }
  return 0;
}

template <int WM>
__device__ __forceinline__ typename ec_valw<WM>::type ec_lookup(const qh_geom& g, const uint64_t* __restrict__ table,
                                                                uint64_t h) {
  typedef typename ec_valw<WM>::type V;
  if (g.wbits < 3) return (V)ec_lookup_narrow_table(g, table, h);
  const uint64_t hh[1] = {h};
  V v[1];
  ec_lookup_wide<1, WM>(g, table, hh, v);
  return v[0];
}

// The four siblings of get_best_alternatives in ONE probe sequence. The position hash takes its low
// bits from the canonical core (bases 1..k-2, qh_spec.h), which the four mers that differ in base 0
// share in whichever orientation their canonical forms lie: they have the same home slot and, probing
// being linear, sit in the same one or two 64-byte lines. One walk from the home slot compares every
// slot with the four tags; it ends when all four are found, at the first empty slot (whatever is still
// missing is absent) or at the reprobe limit -- exactly where the four separate walks would end.
// Needs lsize + shard_bits <= 2(k-2) (the home must not depend on the end bases) and wbits >= 3.
template <int WM>
__device__ __forceinline__ void ec_lookup_siblings(const qh_geom& g, const uint64_t* __restrict__ table, const ec_mer& m,
                                                   typename ec_valw<WM>::type (&w)[4]) {
  typedef typename ec_tagw<WM>::type W;   // tags
  typedef typename ec_valw<WM>::type V;   // values
  const int k = g.k, kc = k - 2;
  const uint32_t cm = (1u << kc) - 1u;
  // the core both strands share, in its smaller orientation, and its hash: the siblings' common low bits
  const uint32_t aL = (m.a0 >> 1) & cm, aR = (m.a1 >> 1) & cm, bL = (m.b0 >> 1) & cm, bR = (m.b1 >> 1) & cm;
  const bool flip_a = qh_planes_greater(aL, aR, bL, bR);
  const bool cores_eq = aL == bL && aR == bR;
  const uint64_t F = qh_feistel3(flip_a ? bL : aL, flip_a ? bR : aR, kc) ^ QH_MIX_Z;
  const uint32_t wmask = (1u << g.wbits) - 1u;
  const int gsize = g.lsize + g.shard_bits;
  const qh_pos p = qh_locate_hash(g, F);
  // sibling y = strand a with base 0 replaced by y, i.e. strand b with base k-1 replaced by 3 - y. Its key is
  // the smaller of the two (kmer.hpp:43), decided at the top base, then by the cores, then at base 0; the
  // key's end bases and the orientation of its core are the hash bits above the core's (qh_mix_pair)
  const uint32_t a_top = ((m.a0 >> (k - 1)) & 1u) | (((m.a1 >> (k - 1)) & 1u) << 1);
  const uint32_t b_low = (m.b0 & 1u) | ((m.b1 & 1u) << 1);
  const W fhi = (W)(F >> gsize);
  const int esh = 2 * kc - gsize;
  W tag[4];
#pragma unroll
  for (uint32_t y = 0; y < 4; ++y) {
    const uint32_t b_top = 3u - y;
    const bool fwd = a_top != b_top ? a_top < b_top : (!cores_eq ? !flip_a : y <= b_low);
    const uint32_t e = fwd ? (y | (a_top << 2)) : (b_low | (b_top << 2));
    const uint32_t fl = fwd ? (uint32_t)flip_a : (uint32_t)(!cores_eq && !flip_a);
    tag[y] = (fhi | ((W)(e | (fl << 4)) << esh)) << g.pbits;
    w[y] = 0;
  }
  if (p.shard != (uint32_t)g.shard_id) return;
  uint64_t gaddr = p.home & ~(uint64_t)(EC_GROUP - 1);
  int pi = -(int)((uint32_t)p.home & (uint32_t)(EC_GROUP - 1));
  // Branch-free per line. A key is stored at the first empty slot of its probe sequence and slots are never
  // emptied, so a slot whose (remainder, probe) fields match a tag at this probe index IS that key, whatever
  // lies before it: only the slots outside [home, home + max_reprobe] must be kept from matching (their
  // difference is forced to 1, which no tag equals: tags have zero probe bits). The walk ends with the
  // line that holds an empty slot or the reprobe limit, or once all four siblings are found.
  for (;;) {
    const ec_group grp = *reinterpret_cast<const ec_group*>(table + gaddr);
    bool stop = pi + EC_GROUP > g.max_reprobe;
    const uint32_t span = (uint32_t)g.max_reprobe;  // valid slots: 0 <= pi + j <= max_reprobe
#pragma unroll
    for (int j = 0; j < EC_GROUP; ++j) {
      const uint64_t sj = grp.s[j];
      const bool valid = (uint32_t)(pi + j) <= span;
      stop |= valid && sj == 0;
      const W t = valid ? (W)(sj >> g.vbits) - (W)(pi + j + 1) : (W)1;
      // (wide form: the slot's index + 1 stands in for its value until the walk is over)
      const V val = (WM == 3 && g.wide) ? (V)(gaddr + (uint64_t)j + 1u) : (V)sj & (V)g.vmask;
#pragma unroll
      for (int y = 0; y < 4; ++y) w[y] = t == tag[y] ? val : w[y];
    }
    // a stored value is never 0 (count >= 1): all four found <=> all four non-zero
    if (stop || (w[0] != 0 && w[1] != 0 && w[2] != 0 && w[3] != 0)) break;
    gaddr = (gaddr & ~(uint64_t)wmask) | (((uint32_t)gaddr + (uint32_t)EC_GROUP) & wmask);
    pi += EC_GROUP;
  }
  if (WM == 3 && g.wide) {
#pragma unroll
    for (int y = 0; y < 4; ++y) w[y] = w[y] ? (V)g.wvals[(uint64_t)w[y] - 1u] : (V)0;
  }
}

// database_query::get_best_alternatives (mer_database.hpp:302-329): one walk for the four siblings (SIB;
// otherwise four lookups advancing together); the fold runs in the order of the forward strand's codes
// like the upstream loop (the candidate with a-strand code y is the forward code y when d > 0 and
// 3 - y when d < 0)
template <int WM, bool SIB>
__device__ __forceinline__ int ec_best_alternatives(const qh_geom& g, const uint64_t* __restrict__ table,
                                                    const ec_mer& m, int d, typename ec_valw<WM>::type (&counts)[4],
                                                    int* ucode, int* level) {
  typedef typename ec_valw<WM>::type W;   // values only here
  const int k = g.k;
  W w[4];
  if (SIB) {
    ec_lookup_siblings<WM>(g, table, m, w);
  } else {
    uint64_t h[4];
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      ec_mer a = m;
      ec_replace0(a, k, (uint32_t)y);
      h[y] = ec_hash(a, k);
    }
    if (g.wbits < 3) {
#pragma unroll
      for (int y = 0; y < 4; ++y) w[y] = (W)ec_lookup_narrow_table(g, table, h[y]);
    } else {
      ec_lookup_wide<4, WM>(g, table, h, w);
    }
  }
  int count = 0, lvl = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    counts[i] = 0;
    const W v = d > 0 ? w[i] : w[3 - i];
    const W first = v >> 1;
    const int second = (int)(v & 1);
    if (first > 0 && second >= lvl) {
      if (second > lvl && count > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < i) counts[j] = 0;
        count = 0;
      }
      counts[i] = first;
      *ucode = i;
      lvl = second;
      ++count;
    }
  }
  *level = lvl;
  return count;
}

template <typename W>
__device__ __forceinline__ W ec_sel4(const W (&c)[4], int i) {
  return i == 0 ? c[0] : i == 1 ? c[1] : i == 2 ? c[2] : c[3];
}

// database_query::get_best_alternatives batched (qh_query_best_alternatives_batch): a mer per lane, the
// four siblings in one walk. fmers are 2-bit packed forward mers; backward[i] selects backward_mer.
template <int WM>
__global__ __launch_bounds__(256) void k_best_alternatives_sib(const uint64_t* __restrict__ table, const qh_geom g,
                                                               const uint64_t* __restrict__ fmers,
                                                               const uint8_t* __restrict__ backward, uint64_t n,
                                                               uint64_t* __restrict__ counts, int32_t* __restrict__ ucode,
                                                               int32_t* __restrict__ level, int32_t* __restrict__ nkept) {
  typedef typename ec_valw<WM>::type W;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t pk = qh_pkey(fmers[i] & g.kmask, g.k);
    const uint32_t km = (uint32_t)((1ull << g.k) - 1);
    const uint32_t f0 = (uint32_t)pk & km, f1 = (uint32_t)(pk >> g.k) & km;
    uint32_t r0, r1;
    qh_rc_planes(f0, f1, g.k, &r0, &r1);
    const int d = backward[i] ? -1 : +1;
    const ec_mer m = d > 0 ? ec_mer{f0, f1, r0, r1} : ec_mer{r0, r1, f0, f1};
    W c[4];
    int uc = 0, lvl = 0;
    const int cnt = ec_best_alternatives<WM, true>(g, table, m, d, c, &uc, &lvl);
#pragma unroll
    for (int a = 0; a < 4; ++a) counts[i * 4 + a] = c[a];
    ucode[i] = uc; level[i] = lvl; nkept[i] = cnt;
  }
}

void qh_launch_best_alternatives(const uint64_t* table, const qh_geom& g, const uint64_t* fmers, const uint8_t* backward,
                                 uint64_t n, uint64_t* counts, int32_t* ucode, int32_t* level, int32_t* nkept,
                                 hipStream_t s) {
  if (!n) return;
  if (!(g.k >= 4 && g.wbits >= 3 && g.lsize + g.shard_bits <= 2 * (g.k - 2))) {
    qh_launch_best_alternatives_generic(table, g, fmers, backward, n, counts, ucode, level, nkept, s);
    return;
  }
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (g.wide)
    hipLaunchKernelGGL(k_best_alternatives_sib<3>, dim3((unsigned)blocks), dim3(256), 0, s, table, g, fmers, backward, n,
                       counts, ucode, level, nkept);
  else if (g.vbits <= 32 && g.rbits + g.pbits <= 32)
    hipLaunchKernelGGL(k_best_alternatives_sib<0>, dim3((unsigned)blocks), dim3(256), 0, s, table, g, fmers, backward, n,
                       counts, ucode, level, nkept);
  else if (g.vbits <= 32)
    hipLaunchKernelGGL(k_best_alternatives_sib<1>, dim3((unsigned)blocks), dim3(256), 0, s, table, g, fmers, backward, n,
                       counts, ucode, level, nkept);
  else
    hipLaunchKernelGGL(k_best_alternatives_sib<2>, dim3((unsigned)blocks), dim3(256), 0, s, table, g, fmers, backward, n,
                       counts, ucode, level, nkept);
}

// ------------------------------------------------------------------ err_log

// entry: pos (24 bits) | type << 24 (0 substitution, 1 truncation) | from << 25 | to << 28 (0..3, 4 = 'N')
#define EC_POS(e) ((int)((e) & 0xFFFFFFu))
struct ec_log {
  uint32_t* e;
  uint32_t n, lwin;
  int d;
};

// err_log::check_nb_error (err_log.hpp:87-95)
__device__ __forceinline__ bool ec_check_nb_error(ec_log& l, int window, int error) {
  if (l.n > 0) {
    const int back = l.d * EC_POS(l.e[l.n - 1]);
    if (back > l.d * window)
      while (back > l.d * EC_POS(l.e[l.lwin]) + window) l.lwin++;
  }
  return (uint64_t)l.n - l.lwin - 1 >= (uint64_t)(uint32_t)error;  // size_t arithmetic upstream
}
__device__ __forceinline__ bool ec_push(ec_log& l, uint32_t entry, int window, int error) {
  l.e[l.n++] = entry;
  return ec_check_nb_error(l, window, error);
}
// truncation as extend() calls it: backward_log::truncation logs pos - 1 in backward-counter
// arithmetic, i.e. raw + 1 (error_correct_reads.hpp:170-172)
__device__ __forceinline__ void ec_truncation(ec_log& l, int pos, int window, int error) {
  ec_push(l, (uint32_t)(l.d > 0 ? pos : pos + 1) | (1u << 24), window, error);
}
// err_log::remove_last_window (err_log.hpp:97-106)
__device__ __forceinline__ int ec_remove_last_window(ec_log& l, int window, int error) {
  if (l.n == 0) return 0;
  const int diff = l.d * (EC_POS(l.e[l.n - 1]) - EC_POS(l.e[l.lwin]));
  l.n = l.lwin;
  l.lwin = 0;
  ec_check_nb_error(l, window, error);
  return diff;
}
// err_log::force_truncate (err_log.hpp:75-83): drops entries with e.pos >= T(pos) in the counter's own order
__device__ __forceinline__ void ec_force_truncate(ec_log& l, int pos, int window, int error) {
  uint32_t w = 0;
  for (uint32_t i = 0; i < l.n; ++i) {
    const uint32_t e = l.e[i];
    if (!(l.d * EC_POS(e) >= l.d * pos)) l.e[w++] = e;
  }
  l.n = w;
  l.lwin = 0;
  ec_check_nb_error(l, window, error);
}

__device__ __forceinline__ uint32_t ec_digits(uint32_t v) {
  return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) +
         (v >= 10000000u);
}
// bytes operator<< (err_log.hpp:111-135) prints for n entries
__device__ __forceinline__ uint32_t ec_log_text_bytes(const uint32_t* e, uint32_t n, uint32_t* subs) {
  uint32_t b = n ? n - 1 : 0;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t x = e[i];
    const bool trunc = (x >> 24) & 1u;
    b += ec_digits(x & 0xFFFFFFu) + 8u;  // ":3_trunc" / ":sub:X-Y"
    *subs += !trunc;
  }
  return b;
}
__device__ __forceinline__ uint8_t* ec_put_uint(uint8_t* w, uint32_t v) {
  const uint32_t nd = ec_digits(v);
  for (uint32_t i = nd; i-- > 0;) { w[i] = (uint8_t)('0' + v % 10u); v /= 10u; }
  return w + nd;
}
__device__ __forceinline__ uint8_t* ec_put_log(uint8_t* w, const uint32_t* e, uint32_t n, int d) {
  const char* const names = "ACGTN";
  for (uint32_t i = 0; i < n; ++i) {
    if (i) *w++ = ' ';
    const uint32_t x = e[i];
    w = ec_put_uint(w, x & 0xFFFFFFu);
    if ((x >> 24) & 1u) {
      *w++ = ':'; *w++ = d > 0 ? '3' : '5'; *w++ = '_'; *w++ = 't'; *w++ = 'r'; *w++ = 'u'; *w++ = 'n'; *w++ = 'c';
    } else {
      *w++ = ':'; *w++ = 's'; *w++ = 'u'; *w++ = 'b'; *w++ = ':';
      *w++ = (uint8_t)names[(x >> 25) & 7u]; *w++ = '-'; *w++ = (uint8_t)names[(x >> 28) & 7u];
    }
  }
  return w;
}

__device__ __forceinline__ uint32_t ec_error_text_bytes(uint32_t status) {
  // "Contaminated read" / "No high quality mer" / "Entire read is an homopolymer" (error_correct_reads.cc:646-648)
  return status == EC_ST_CONTAMINANT ? 17u : status == EC_ST_NO_START ? 19u : 29u;
}

// ------------------------------------------------------------------ the corrector

// is_contaminant (error_correct_reads.cc:692-708) for the hash of a canonical mer
template <int WM>
__device__ __forceinline__ bool ec_is_contaminant(const qh_ec_dev& P, const uint64_t* __restrict__ ctable, uint64_t h) {
  if (P.cbits_mask) {
    const uint32_t i = (uint32_t)h & P.cbits_mask;
    if (!((P.cbits[i >> 5] >> (i & 31u)) & 1u)) return false;
  }
  return ec_lookup<WM>(P.cg, ctable, h) != 0;
}

__global__ __launch_bounds__(256) void k_ec_contaminant_bits(const uint64_t* __restrict__ ctable, const qh_geom cg, uint64_t nslots,
                                                             uint32_t* __restrict__ bits, uint32_t mask) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t s = ctable[i];
    if (s) {
      const uint32_t b = (uint32_t)qh_slot_hash(cg, s, i) & mask;
      atomicOr(&bits[b >> 5], 1u << (b & 31u));
    }
  }
}

// check_contaminant (error_correct_reads.cc:346-357)
template <int WM>
__device__ __forceinline__ int ec_check_contaminant(const qh_ec_dev& P, const uint64_t* __restrict__ ctable, const ec_mer& m,
                                                    ec_log& log, int cpos) {
  if (ctable && ec_is_contaminant<WM>(P, ctable, ec_hash(m, P.g.k))) {
    if (P.trim_contaminant) { ec_truncation(log, cpos, P.window, P.error); return EC_TRUNCATE; }
    return EC_ERROR;
  }
  return EC_OK;
}

// log_substitution (error_correct_reads.cc:360-379); from / to are codes of the forward strand
template <int WM>
__device__ __forceinline__ int ec_log_substitution(const qh_ec_dev& P, const uint64_t* __restrict__ ctable, ec_mer& m, int d,
                                                   int& out, ec_log& log, int cpos, int from, int to) {
  if (from == to) return EC_OK;
  ec_replace0(m, P.g.k, (uint32_t)(d > 0 ? to : 3 - to));
  const int cc = ec_check_contaminant<WM>(P, ctable, m, log, cpos);
  if (cc != EC_OK) return cc;
  const uint32_t entry = (uint32_t)cpos | ((uint32_t)(from >= 0 ? from : 4) << 25) | ((uint32_t)(to >= 0 ? to : 4) << 28);
  if (ec_push(log, entry, P.window, P.error)) {
    const int diff = ec_remove_last_window(log, P.window, P.error);
    out -= d * diff;
    ec_truncation(log, cpos - d * diff, P.window, P.error);
    return EC_TRUNCATE;
  }
  return EC_OK;
}

__device__ __forceinline__ double ec_poisson_term(double lambda, unsigned i) {  // error_correct_reads.cc:53-61
  const double facts[11] = {1, 1, 2, 6, 24, 120, 720, 5040, 40320, 362880, 3628800};
  const double tau = 6.283185307179583;
  if (i < 11) return exp(-lambda) * pow(lambda, (double)i) / facts[i];
  return exp(-lambda + i) * pow(lambda / i, (double)i) / sqrt(tau * i);
}

// The read's characters and the corrected bases go through 16-byte register buffers: a lane touches
// one base per step and 100-300 steps pass between the first and the last byte of a 64-byte line, far
// longer than a line survives in L2 under the table traffic. Byte-wise, every step re-fetched the sequence
// line (1.1 of the 2.4 HBM lines per base) and the byte stores reached HBM as partial-line writes
// (5.7 GB written for 200 MB of bases).
struct ec_rbuf {  // one aligned 16-byte block of the chunk
  uint4 w;
  uint32_t blk;
};
__device__ __forceinline__ uint8_t ec_read(ec_rbuf& r, const uint8_t* __restrict__ buf, size_t buf_len, uint32_t off) {
  const uint32_t blk = off & ~15u;
  if (blk != r.blk) { r.w = load16_guarded(buf, blk, buf_len); r.blk = blk; }
  const uint32_t wi = (off >> 2) & 3u;
  const uint32_t x = wi == 0 ? r.w.x : wi == 1 ? r.w.y : wi == 2 ? r.w.z : r.w.w;
  return (uint8_t)(x >> (8u * (off & 3u)));
}
struct ec_wbuf {  // bytes written into one aligned 16-byte block of the output, not yet stored
  uint32_t w[4];
  uint32_t blk, mask;
};
__device__ __forceinline__ void ec_flush(ec_wbuf& b, uint8_t* __restrict__ obuf) {
  if (b.mask == 0xFFFFu) {
    *reinterpret_cast<uint4*>(obuf + b.blk) = make_uint4(b.w[0], b.w[1], b.w[2], b.w[3]);
  } else if (b.mask) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if ((b.mask >> i) & 1u) obuf[b.blk + i] = (uint8_t)(b.w[i >> 2] >> (8 * (i & 3)));
  }
  b.mask = 0;
}
__device__ __forceinline__ void ec_write(ec_wbuf& b, uint8_t* __restrict__ obuf, uint32_t off, uint8_t c) {
  const uint32_t blk = off & ~15u;
  if (blk != b.blk) { ec_flush(b, obuf); b.blk = blk; }
  const uint32_t wi = (off >> 2) & 3u, sh = 8u * (off & 3u);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (wi == (uint32_t)i) b.w[i] = (b.w[i] & ~(0xFFu << sh)) | ((uint32_t)c << sh);
  b.mask |= 1u << (off & 15u);
}

template <int WM, bool SIB>
__global__ __launch_bounds__(EC_BLOCK) __attribute__((amdgpu_waves_per_eu(EC_WAVES, EC_WAVES))) void k_correct_reads(const uint8_t* __restrict__ buf, size_t buf_len,
                                                            const uint4* __restrict__ recs,
                                                            const uint2* __restrict__ hdr,
                                                            const qh_meta* __restrict__ meta,
                                                            const qh_ec_dev* __restrict__ Pp,
                                                            const uint64_t* __restrict__ table,
                                                            const uint64_t* __restrict__ ctable,
                                                            const uint32_t* __restrict__ ptab,
                                                            uint8_t* __restrict__ obuf, uint32_t* __restrict__ logs,
                                                            uint32_t* __restrict__ res,
                                                            qh_ec_totals* __restrict__ tot) {
  typedef typename ec_valw<WM>::type W;   // values and counts; tags stay inside the lookups
  if (meta->format_err) return;
  const qh_ec_dev& P = *Pp;
  const uint32_t nrec = meta->nrec;
  const int k = P.g.k;
  const uint32_t km = (uint32_t)((1ull << k) - 1);
  uint32_t t_corrected = 0, t_skipped = 0, t_subs = 0, t_truncs = 0, t_reads = 0;
  unsigned long long t_in = 0, t_out = 0;

  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nrec; r += gridDim.x * blockDim.x) {
    const uint4 rc = recs[r];
    uint8_t* __restrict__ o = obuf + rc.x;
    ec_rbuf rb;
    rb.blk = 0xFFFFFFFFu;
    ec_wbuf wb;
    wb.blk = 0xFFFFFFFFu; wb.mask = 0;
    const int len = (int)rc.y;
    // strict 4-line chunks: consecutive sequences are >= 2 len + 6 bytes apart, so the word range
    // [seq_off + 4 r, + len + 4) is private to the read; host-parsed chunks carry the offset in .w
    const uint32_t log_off = hdr ? rc.w : rc.x + 4u * r;
    uint32_t hlen;
    if (hdr) hlen = hdr[r].y;
    else {
      const uint32_t line = r ? recs[r - 1].z + recs[r - 1].y + 1u : 0u;
      hlen = rc.x - 1u - (line + 1u);
    }

    uint32_t status = EC_ST_OK;
    ec_mer mer = {0, 0, 0, 0};
    int in = P.skip;
    // ---- find_starting_mer (error_correct_reads.cc:609-643); the raw characters are the output here
    bool anchored = false;
    while (in < len && !anchored && status == EC_ST_OK) {
      int i = 0;
      while (in < len && i < k) {
        const uint8_t c = ec_read(rb, buf, buf_len, rc.x + (uint32_t)in);
        ec_write(wb, obuf, rc.x + (uint32_t)in, c);
        ++in;
        const int x = qh_code(c);
        if (x >= 0) { ec_shift(mer, k, km, (uint32_t)x); ++i; }
        else i = 0;  // upstream tests != -1 (kmer.hpp:15-22): see the oracle's header for the one deviation
      }
      int found = 0;
      while (in < len) {
        const uint64_t h = ec_hash(mer, k);
        const bool contaminated = ctable && ec_is_contaminant<WM>(P, ctable, h);
        if (contaminated && !P.trim_contaminant) { status = EC_ST_CONTAMINANT; break; }
        if (!contaminated) {
          const W v = ec_lookup<WM>(P.g, table, h);
          const W val = (v & 1) ? v >> 1 : 0;  // get_val
          found = (int)val >= P.anchor ? found + 1 : 0;
          if (found >= P.good) { anchored = true; break; }
        }
        const uint8_t c = ec_read(rb, buf, buf_len, rc.x + (uint32_t)in);
        ec_write(wb, obuf, rc.x + (uint32_t)in, c);
        ++in;
        const int x = qh_code(c);
        if (x < 0) break;
        ec_shift(mer, k, km, (uint32_t)x);
      }
    }
    if (!anchored && status == EC_ST_OK) status = EC_ST_NO_START;

    int start_out = 0, end_out = 0;
    uint32_t nf = 0, nb = 0, boff = log_off;
    if (status == EC_ST_OK) {
      // ---- extend forward, then backward (error_correct_reads.cc:279-316, 386-565) in one loop
      const ec_mer anchor_mer = mer;
      const int start_off = in;
      uint32_t prev0;
      {
        const W v = ec_lookup<WM>(P.g, table, ec_hash(mer, k));
        prev0 = (uint32_t)((v & 1) ? v >> 1 : 0);
      }
      int d = +1;
      ec_log log = {logs + log_off, 0u, 0u, +1};
      int ip = start_off, pos = start_off, end = len, out = start_off;
      uint32_t prev_count = prev0;
      // the character under the head is loaded one step ahead: it never waits behind the lookups
      uint8_t nbase = ip != end ? ec_read(rb, buf, buf_len, rc.x + (uint32_t)ip) : (uint8_t)0;
      // A base normally costs one iteration: shift, one sibling walk, decide. Where several alternatives
      // survive (error_correct_reads.cc:464-560) each candidate needs the walk of ITS continuation; instead of
      // a second walk nested in the step -- which the whole wave would execute whenever one lane needs it,
      // about every other step -- the lane stays on the base (in_cont) and spends its next iterations on
      // the candidates through the SAME walk code the other lanes use for their next bases.
      bool in_cont = false;
      uint32_t qm = 0, succ = 0, cwcb = 0;  // candidates left / that continue / that continue with the read's next base
      W s_counts[4] = {0, 0, 0, 0};         // the base's counts while its candidates are looked up
      int s_level = 0, check_code = 0, ori_code = 0, cpos = 0, cur_i = 0;
      for (;;) {
        bool fin = false, emit = false, finalize = false, lookup = false;
        ec_mer lm = mer;
        if (!in_cont) {
          fin = ip == end;
          if (!fin) {
            const uint8_t base = nbase;
            nbase = ip + d != end ? ec_read(rb, buf, buf_len, rc.x + (uint32_t)(ip + d)) : (uint8_t)0;
            if (base == '\n') { ip += d; continue; }
            cpos = pos;
            pos += d;
            ori_code = qh_code(base);
            const uint32_t x = ori_code >= 0 ? (uint32_t)ori_code : 0u;
            ec_shift(mer, k, km, d > 0 ? x : 3u - x);
            if (ori_code >= 0 && ctable) {
              const int cc = ec_check_contaminant<WM>(P, ctable, mer, log, cpos);
              if (cc == EC_TRUNCATE) fin = true;
              else if (cc == EC_ERROR) { status = EC_ST_CONTAMINANT; break; }
            }
            lm = mer;
            lookup = !fin;
          }
        } else {  // the continuation of candidate cur_i: nmer.replace(0, i); nmer.shift(0) (:492-496)
          cur_i = __ffs((int)qm) - 1;
          qm &= qm - 1u;
          ec_replace0(lm, k, (uint32_t)(d > 0 ? cur_i : 3 - cur_i));
          ec_shift(lm, k, km, d > 0 ? 0u : 3u);
          lookup = true;
        }
        const bool stepping = lookup || fin;  // an iteration that works on a base (fin: cut short before its lookup)
        W counts[4] = {0, 0, 0, 0};
        int ucode = 0, level = 0, count = 0;
        if (lookup) count = ec_best_alternatives<WM, SIB>(P.g, table, lm, d, counts, &ucode, &level);
        if (lookup && in_cont) {
          check_code = cur_i;
          if (count > 0 && level >= s_level) {  // :502-506
            const int rn = ip + d != end ? qh_code(nbase) : -1;
            if (rn >= 0 && ec_sel4(counts, rn) > 0) cwcb |= 1u << cur_i;
            succ |= 1u << cur_i;
          }
          if (!qm) { in_cont = false; finalize = true; }
        } else if (lookup) {
          if (count == 1 && ucode == ori_code) {  // the read's base is the one continuation: the common case
            prev_count = (uint32_t)ec_sel4(counts, ucode);
            emit = true;
          } else if (count == 0) {  // no continuation whatsoever
            ec_truncation(log, cpos, P.window, P.error);
            fin = true;
          } else if (count == 1) {  // one continuation: a substitution
            prev_count = (uint32_t)ec_sel4(counts, ucode);
            const int lc = ec_log_substitution<WM>(P, ctable, mer, d, out, log, cpos, ori_code, ucode);
            if (lc == EC_TRUNCATE) fin = true;
            else if (lc == EC_ERROR) { status = EC_ST_CONTAMINANT; break; }
            else emit = true;
          } else {
            bool multi = true;
            if (ori_code >= 0) {
              const uint64_t co = ec_sel4(counts, ori_code);
              if (co > (uint64_t)P.min_count) {
                if (co >= (uint32_t)P.cutoff || (int)(int8_t)buf[rc.z + (uint32_t)ip] >= (int)(int8_t)P.qual_cutoff) {
                  emit = true; multi = false;
                } else {
                  const uint64_t sum = (uint64_t)counts[0] + counts[1] + counts[2] + counts[3];
                  bool keep;
                  if (P.ptab_stride) {
                    const uint64_t bit = co * (uint64_t)P.ptab_stride + sum;
                    keep = (ptab[bit >> 5] >> (bit & 31)) & 1u;
                  } else {
                    keep = ec_poisson_term((double)sum * P.collision_prob, (unsigned)co) < P.poisson_threshold;
                  }
                  if (keep) { emit = true; multi = false; }
                }
              } else if (level == 0 && co == 0) {  // an error and all alternatives are low quality
                ec_truncation(log, cpos, P.window, P.error);
                fin = true; multi = false;
              }
            } else if (level == 0) {  // N and all alternatives are low quality
              ec_truncation(log, cpos, P.window, P.error);
              fin = true; multi = false;
            }
            if (multi) {  // several possible substitutions: look the candidates' continuations up first
              check_code = ori_code;
              succ = 0; cwcb = 0; s_level = level;
              qm = 0;
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                s_counts[i] = counts[i];
                qm |= (uint32_t)((uint64_t)counts[i] > (uint64_t)P.min_count) << i;  // :488-489
              }
              if (qm) in_cont = true;
              else finalize = true;
            }
          }
        }
        if (finalize) {  // :509-560 once every candidate has been looked up
          if (succ) {
            check_code = -1;
            const int rn = ip + d != end ? qh_code(nbase) : -1;
            const uint32_t pc = prev_count <= (uint64_t)P.min_count ? 0xFFFFFFFFu : prev_count;
            int min_diff = INT_MAX;
            uint32_t cand = 0;
            unsigned ncand = 0;
            long long ad[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const long long cc = (succ >> i) & 1u ? (long long)s_counts[i] : 0ll;  // cont_counts[i]
              ad[i] = llabs(cc - (long long)pc);
              if (cc > 0) {
                const int dd = (int)ad[i];  // upstream's (int) cast, :521
                if (dd < min_diff) min_diff = dd;
              }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (ad[i] == (long long)min_diff) { cand |= 1u << i; ++ncand; check_code = i; }  // :526
            if (ncand > 1 && rn >= 0) {
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if ((cand >> i) & 1u) {
                  if (!((cwcb >> i) & 1u)) --ncand;
                  else check_code = i;
                }
            }
            if (ncand != 1) check_code = -1;
            if (check_code >= 0) {
              const int lc = ec_log_substitution<WM>(P, ctable, mer, d, out, log, cpos, ori_code, check_code);
              if (lc == EC_TRUNCATE) fin = true;
              else if (lc == EC_ERROR) { status = EC_ST_CONTAMINANT; break; }
            }
          }
          if (!fin) {
            if (ori_code < 0 && check_code < 0) {  // invalid base and no good substitution
              ec_truncation(log, cpos, P.window, P.error);
              fin = true;
            } else {
              emit = true;
            }
          }
        }
        if (emit) {  // *out++ = mer.base(0)
          const uint32_t c0 = ec_code0(mer);
          ec_write(wb, obuf, rc.x + (uint32_t)out, (uint8_t)(0x54474341u >> (8u * (d > 0 ? c0 : 3u - c0)))  /* "ACGT"[code] */);
          out += d;
        }
        if (lookup && !in_cont) ip += d;  // the base is done
        (void)stepping;
        if (fin) {
          if (d > 0) {
            end_out = out;
            nf = log.n;
            d = -1;
            boff = log_off + nf + 1;  // one word left free: homo_trim may append a truncation to the forward log
            log.e = logs + boff;
            log.n = 0; log.lwin = 0; log.d = -1;
            mer.a0 = anchor_mer.b0; mer.a1 = anchor_mer.b1; mer.b0 = anchor_mer.a0; mer.b1 = anchor_mer.a1;
            ip = start_off - k - 1; pos = ip; end = -1; out = ip;
            prev_count = prev0;
            nbase = ip != end ? ec_read(rb, buf, buf_len, rc.x + (uint32_t)ip) : (uint8_t)0;
          } else {
            start_out = out + 1;
            nb = log.n;
            break;
          }
        }
      }
    }

    ec_flush(wb, obuf);
    uint32_t* __restrict__ lg = logs + log_off;
    // ---- homo_trim (error_correct_reads.cc:567-597)
    if (status == EC_ST_OK && P.do_homo_trim) {
      int max_homo_score = INT_MIN, max_pos = INT_MIN, homo_score = 0;
      int ptr = end_out - 1;
      int pbase = qh_code(o[ptr]);
      for (--ptr; ptr >= start_out; --ptr) {
        const int cbase = qh_code(o[ptr]);
        homo_score += ((pbase == cbase) << 1) - 1;
        pbase = cbase;
        if (homo_score > max_homo_score) { max_homo_score = homo_score; max_pos = ptr; }
      }
      if (max_homo_score >= P.homo_trim) {
        if (max_pos < start_out) status = EC_ST_HOMOPOLYMER;
        else {
          ec_log fl = {lg, nf, 0u, +1}, bl = {logs + boff, nb, 0u, -1};
          ec_force_truncate(fl, max_pos, P.window, P.error);
          ec_force_truncate(bl, max_pos, P.window, P.error);
          ec_push(fl, (uint32_t)max_pos | (1u << 24), P.window, P.error);  // fwd_log.truncation(max_pos)
          nf = fl.n; nb = bl.n;  // the forward log grew by at most the one word left free
          end_out = max_pos;
        }
      }
    }

    // ---- sizes of this read's text in both streams
    uint32_t fa_bytes = 0, log_bytes = 0, subs = 0;
    ++t_reads;
    t_in += (unsigned long long)len;
    if (status == EC_ST_OK) {
      const uint32_t fb = ec_log_text_bytes(lg, nf, &subs), bb = ec_log_text_bytes(logs + boff, nb, &subs);
      const uint32_t slen = end_out > start_out ? (uint32_t)(end_out - start_out) : 0u;
      fa_bytes = 1u + hlen + 1u + fb + 1u + bb + 1u + slen + 1u;
      ++t_corrected;
      t_subs += subs;
      t_truncs += nf + nb - subs;
      t_out += slen;
    } else {
      log_bytes = 8u + hlen + 2u + ec_error_text_bytes(status) + 1u;  // "Skipped " header ": " reason "\n"
      if (P.no_discard) fa_bytes = 1u + hlen + 3u;                     // ">" header "\nN\n"
      ++t_skipped;
    }
    uint32_t* __restrict__ rr = res + (size_t)r * EC_RES_WORDS;
    reinterpret_cast<uint4*>(rr)[0] = make_uint4(status, (uint32_t)start_out, (uint32_t)end_out, nf);
    reinterpret_cast<uint4*>(rr)[1] = make_uint4(nb, boff, fa_bytes, log_bytes);
  }

  // chunk statistics: one atomic per wave and counter
  t_reads = wave_sum_u32(t_reads); t_corrected = wave_sum_u32(t_corrected); t_skipped = wave_sum_u32(t_skipped);
  t_subs = wave_sum_u32(t_subs); t_truncs = wave_sum_u32(t_truncs);
  uint32_t in_lo = wave_sum_u32((uint32_t)(t_in & 0xFFFFFu)), in_hi = wave_sum_u32((uint32_t)(t_in >> 20));
  uint32_t out_lo = wave_sum_u32((uint32_t)(t_out & 0xFFFFFu)), out_hi = wave_sum_u32((uint32_t)(t_out >> 20));
  if ((threadIdx.x & 63) == 0 && t_reads) {
    atomicAdd(&tot->reads, (unsigned long long)t_reads);
    atomicAdd(&tot->corrected, (unsigned long long)t_corrected);
    atomicAdd(&tot->skipped, (unsigned long long)t_skipped);
    atomicAdd(&tot->substitutions, (unsigned long long)t_subs);
    atomicAdd(&tot->truncations, (unsigned long long)t_truncs);
    atomicAdd(&tot->bases_in, ((unsigned long long)in_hi << 20) + in_lo);
    atomicAdd(&tot->bases_out, ((unsigned long long)out_hi << 20) + out_lo);
  }
}

// ------------------------------------------------------------------ output text

// per block of EC_FMT_BLOCK reads: bytes of both streams
__global__ __launch_bounds__(EC_FMT_BLOCK) void k_ec_block_sums(const uint32_t* __restrict__ res,
                                                                const qh_meta* __restrict__ meta,
                                                                unsigned long long* __restrict__ bsums) {
  __shared__ unsigned long long sh[2][EC_FMT_BLOCK / 64];
  const uint32_t nrec = meta->format_err ? 0u : meta->nrec;
  // the grid is sized from the chunk length (~13 times the blocks that hold records); nobody reads the sums of
  // the others (k_ec_scan_blocks and k_ec_format stop at nrec)
  if (blockIdx.x * EC_FMT_BLOCK >= nrec) return;
  const uint32_t r = blockIdx.x * EC_FMT_BLOCK + threadIdx.x;
  uint32_t a = 0, b = 0;
  if (r < nrec) {
    const uint2 ab = *reinterpret_cast<const uint2*>(res + (size_t)r * EC_RES_WORDS + 6);
    a = ab.x; b = ab.y;
  }
  // per-read sizes are < 2^32 / 64 in any chunk the library accepts (chunk <= 4 GB), sum in 64 bits anyway
  unsigned long long sa = a, sb = b;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = sa; sh[1][threadIdx.x >> 6] = sb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long ta = 0, tb = 0;
    for (int w = 0; w < EC_FMT_BLOCK / 64; ++w) { ta += sh[0][w]; tb += sh[1][w]; }
    bsums[2 * (size_t)blockIdx.x] = ta;
    bsums[2 * (size_t)blockIdx.x + 1] = tb;
  }
}

// exclusive scan of the block sums (one workgroup), totals
__global__ __launch_bounds__(1024) void k_ec_scan_blocks(unsigned long long* __restrict__ bsums, uint32_t nblocks_cap,
                                                         const qh_meta* __restrict__ meta,
                                                         qh_ec_totals* __restrict__ tot) {
  __shared__ unsigned long long pa[1024], pb[1024];
  // only the blocks that hold records (the capacity is sized from the chunk length, ~13 times more)
  uint32_t nblocks = meta->format_err ? 0u : (meta->nrec + EC_FMT_BLOCK - 1) / EC_FMT_BLOCK;
  if (nblocks > nblocks_cap) nblocks = nblocks_cap;
  const uint32_t per = (nblocks + 1023) / 1024;
  const uint32_t b0 = threadIdx.x * per;
  unsigned long long sa = 0, sb = 0;
  for (uint32_t i = b0; i < b0 + per && i < nblocks; ++i) { sa += bsums[2 * (size_t)i]; sb += bsums[2 * (size_t)i + 1]; }
  pa[threadIdx.x] = sa; pb[threadIdx.x] = sb;
  __syncthreads();
  for (uint32_t o = 1; o < 1024; o <<= 1) {
    const unsigned long long xa = threadIdx.x >= o ? pa[threadIdx.x - o] : 0, xb = threadIdx.x >= o ? pb[threadIdx.x - o] : 0;
    __syncthreads();
    pa[threadIdx.x] += xa; pb[threadIdx.x] += xb;
    __syncthreads();
  }
  unsigned long long ra = pa[threadIdx.x] - sa, rb = pb[threadIdx.x] - sb;
  for (uint32_t i = b0; i < b0 + per && i < nblocks; ++i) {
    const unsigned long long xa = bsums[2 * (size_t)i], xb = bsums[2 * (size_t)i + 1];
    bsums[2 * (size_t)i] = ra; bsums[2 * (size_t)i + 1] = rb;
    ra += xa; rb += xb;
  }
  if (threadIdx.x == 1023) {
    tot->fa_bytes = pa[1023];
    tot->log_bytes = pb[1023];
    tot->overflow = (pa[1023] > tot->fa_cap ? 1u : 0u) | (pb[1023] > tot->log_cap ? 2u : 0u);
  }
}

// Every read writes its text: error_correct_reads.cc:272-275 (skipped) and :334-336 (corrected).
// A block of 256 reads assembles its stretch of the .fa stream in LDS -- every lane the header and the
// logs of its read, then each wave the corrected bases of its 64 reads one read at a time (coalesced
// loads instead of 64 lanes reading 64 different lines) -- and streams it out in 16-byte units. A block
// whose text does not fit (very long reads or logs) writes straight to global memory instead.
#define EC_FMT_LDS 57344u
__global__ __launch_bounds__(EC_FMT_BLOCK) void k_ec_format(const uint8_t* __restrict__ buf,
                                                            const uint4* __restrict__ recs,
                                                            const uint2* __restrict__ hdr,
                                                            const qh_meta* __restrict__ meta,
                                                            const uint8_t* __restrict__ obuf,
                                                            const uint32_t* __restrict__ logs,
                                                            const uint32_t* __restrict__ res,
                                                            const unsigned long long* __restrict__ bsums,
                                                            const qh_ec_totals* __restrict__ tot,
                                                            uint8_t* __restrict__ fa, uint8_t* __restrict__ lgout) {
  __shared__ uint32_t sa[EC_FMT_BLOCK], sb[EC_FMT_BLOCK];
  __shared__ uint32_t cp_src[EC_FMT_BLOCK], cp_dst[EC_FMT_BLOCK], cp_n[EC_FMT_BLOCK];
  __shared__ __align__(16) uint8_t text[EC_FMT_LDS + 32];
  if (meta->format_err) return;
  const uint32_t nrec = meta->nrec;
  const uint32_t tid = threadIdx.x;
  const uint32_t r = blockIdx.x * EC_FMT_BLOCK + tid;
  if (blockIdx.x * EC_FMT_BLOCK >= nrec) return;
  uint4 r0 = make_uint4(0, 0, 0, 0), r1 = make_uint4(0, 0, 0, 0);
  if (r < nrec) {
    r0 = reinterpret_cast<const uint4*>(res + (size_t)r * EC_RES_WORDS)[0];
    r1 = reinterpret_cast<const uint4*>(res + (size_t)r * EC_RES_WORDS)[1];
  }
  sa[tid] = r1.z; sb[tid] = r1.w;
  __syncthreads();
  for (uint32_t o = 1; o < EC_FMT_BLOCK; o <<= 1) {
    const uint32_t xa = tid >= o ? sa[tid - o] : 0, xb = tid >= o ? sb[tid - o] : 0;
    __syncthreads();
    sa[tid] += xa; sb[tid] += xb;
    __syncthreads();
  }
  const uint32_t block_bytes = sa[EC_FMT_BLOCK - 1];
  const unsigned long long blk_off = bsums[2 * (size_t)blockIdx.x];
  const bool fa_ok = !(tot->overflow & 1u);
  const bool staged = block_bytes <= EC_FMT_LDS;
  const uint32_t mis = (uint32_t)(blk_off & 15u);  // LDS copy congruent to the global offset modulo 16
  const uint32_t loff = sa[tid] - r1.z;            // this read's offset inside the block's text
  uint8_t* const wbase = staged ? text + mis : fa + blk_off;
  cp_n[tid] = 0;
  if (r < nrec) {
    const unsigned long long lg_off = bsums[2 * (size_t)blockIdx.x + 1] + sb[tid] - r1.w;
    const uint4 rc = recs[r];
    uint32_t hoff, hlen;
    if (hdr) { hoff = hdr[r].x; hlen = hdr[r].y; }
    else {
      const uint32_t line = r ? recs[r - 1].z + recs[r - 1].y + 1u : 0u;
      hoff = line + 1u;
      hlen = rc.x - 1u - hoff;
    }
    const uint8_t* __restrict__ h = buf + hoff;
    const uint32_t status = r0.x;
    if (status == EC_ST_OK) {
      if (fa_ok) {
        uint8_t* w = wbase + loff;
        *w++ = '>';
        for (uint32_t i = 0; i < hlen; ++i) *w++ = h[i];
        *w++ = ' ';
        w = ec_put_log(w, logs + (hdr ? rc.w : rc.x + 4u * r), r0.w, +1);
        *w++ = ' ';
        w = ec_put_log(w, logs + r1.y, r1.x, -1);
        *w++ = '\n';
        const uint32_t n = (int)r0.z > (int)r0.y ? r0.z - r0.y : 0u;
        if (staged) {
          cp_src[tid] = rc.x + r0.y;
          cp_dst[tid] = (uint32_t)(w - wbase);
          cp_n[tid] = n;
        } else {
          const uint8_t* __restrict__ o = obuf + rc.x + r0.y;
          for (uint32_t i = 0; i < n; ++i) w[i] = o[i];
        }
        w[n] = '\n';
      }
    } else {
      if (!(tot->overflow & 2u)) {
        uint8_t* w = lgout + lg_off;
        const char* const pre = "Skipped ";
        for (int i = 0; i < 8; ++i) *w++ = (uint8_t)pre[i];
        for (uint32_t i = 0; i < hlen; ++i) *w++ = h[i];
        *w++ = ':'; *w++ = ' ';
        const char* const msg = status == EC_ST_CONTAMINANT ? "Contaminated read"
                                : status == EC_ST_NO_START  ? "No high quality mer"
                                                            : "Entire read is an homopolymer";
        const uint32_t ml = ec_error_text_bytes(status);
        for (uint32_t i = 0; i < ml; ++i) *w++ = (uint8_t)msg[i];
        *w++ = '\n';
      }
      if (r1.z && fa_ok) {  // -d: ">" header "\nN\n"
        uint8_t* w = wbase + loff;
        *w++ = '>';
        for (uint32_t i = 0; i < hlen; ++i) *w++ = h[i];
        *w++ = '\n'; *w++ = 'N'; *w++ = '\n';
      }
    }
  }
  if (!staged || !fa_ok) return;  // uniform over the block
  __syncthreads();
  // corrected bases: wave w copies the sequences of reads 64 w .. 64 w + 63, 64 bytes per step
  const uint32_t lane = tid & 63u, w0 = tid & ~63u;
  // (eight reads at a time, so that their loads are in flight together: one read after the other, each load
  // waited for, made this loop the whole kernel's time)
  for (uint32_t i0 = 0; i0 < 64; i0 += 8) {
    uint32_t n[8], a[8], b[8];
    const uint8_t* src[8];
    uint8_t* dst[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      n[u] = cp_n[w0 + i0 + u];
      src[u] = obuf + cp_src[w0 + i0 + u];
      dst[u] = text + mis + cp_dst[w0 + i0 + u];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = lane < n[u] ? src[u][lane] : 0u;
      b[u] = lane + 64u < n[u] ? src[u][lane + 64u] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (lane < n[u]) dst[u][lane] = (uint8_t)a[u];
      if (lane + 64u < n[u]) dst[u][lane + 64u] = (uint8_t)b[u];
      for (uint32_t x = lane + 128u; x < n[u]; x += 64) dst[u][x] = src[u][x];  // reads of more than 128 bases
    }
  }
  __syncthreads();
  // stream the block's text out: whole 16-byte units as one store, the ragged ends byte by byte
  const uint32_t end = mis + block_bytes, units = (end + 15u) / 16u;
  uint8_t* const gbase = fa + (blk_off - mis);  // 16-byte aligned: fa comes from hipMalloc
  for (uint32_t u = tid; u < units; u += EC_FMT_BLOCK) {
    const uint32_t b0 = u * 16u;
    if (b0 >= mis && b0 + 16u <= end) {
      *reinterpret_cast<uint4*>(gbase + b0) = *reinterpret_cast<const uint4*>(text + b0);
    } else {
      for (uint32_t b = b0 < mis ? mis : b0; b < b0 + 16u && b < end; ++b) gbase[b] = text[b];
    }
  }
}

// ------------------------------------------------------------------ host side

struct qh_corrector {
  int device = 0;
  hipStream_t stream = nullptr;
  qh_query* db = nullptr;
  qh_query* contaminant = nullptr;
  qh_correct_params params;
  qh_ec_dev P;
  qh_ec_dev* d_P = nullptr;    // the same block in device memory (the kernel reads it through the scalar cache)
  uint32_t* d_cbits = nullptr; // bit filter in front of the contaminant table (qh_ec_dev::cbits)
  int word_mode = 2;           // 0: value and tag fields fit 32 bits; 1: 64-bit tags, 32-bit values; 2: all 64-bit
  bool siblings = false;       // the home slot does not depend on the end bases: one walk per get_best_alternatives
  uint32_t* d_ptab = nullptr;
  uint64_t* own_table = nullptr;  // the database re-hashed into a sparser table for the walks (corrector_init)
  // chunk state
  uint8_t* d_chunk = nullptr;  size_t chunk_cap = 0;   // staging of host-fed chunks
  uint4* d_recs = nullptr;     size_t rec_cap = 0;
  uint2* d_hdr = nullptr;      size_t hdr_cap = 0;
  uint32_t* d_tiles = nullptr; size_t tile_cap = 0;
  ushort4* d_nlplane = nullptr; size_t nlplane_cap = 0;
  uint8_t* d_obuf = nullptr;   size_t obuf_cap = 0;
  uint32_t* d_logs = nullptr;  size_t logs_cap = 0;
  uint32_t* d_res = nullptr;   size_t res_cap = 0;
  unsigned long long* d_bsums = nullptr; size_t bsums_cap = 0;
  uint8_t* d_fa = nullptr;     size_t fa_cap = 0;
  uint8_t* d_log = nullptr;    size_t log_cap = 0;
  // pinned, two of each: fetches alternate between them, so what one fetch returned stays valid through the next
  // correct + fetch (a caller may write chunk i out while chunk i+1 is corrected)
  char* h_fa[2] = {nullptr, nullptr};   size_t h_fa_cap[2] = {0, 0};
  char* h_log[2] = {nullptr, nullptr};  size_t h_log_cap[2] = {0, 0};
  int h_out = 0;
  qh_meta* d_meta = nullptr;   qh_meta* h_meta = nullptr;
  qh_ec_totals* d_tot = nullptr; qh_ec_totals* h_tot = nullptr;
  hipEvent_t ev[4] = {};
  // the chunk in flight (for a re-run of the format kernel and for fetch)
  const uint8_t* cur_buf = nullptr;
  const uint2* cur_hdr = nullptr;
  uint32_t cur_rec_cap = 0;
  bool pending = false;
};

static int ensure_dev(void** p, size_t* cap, size_t need, const char* what) { return qh_ensure_device(p, cap, need, what); }

static int ensure_pinned(char** p, size_t* cap, size_t need) {
  if (*cap >= need && *p) return QH_OK;
  if (*p) { (void)hipHostFree(*p); *p = nullptr; *cap = 0; }
  const size_t want = need + need / 8 + 4096;
  hipError_t e = hipHostMalloc((void**)p, want, hipHostMallocDefault);
  if (e != hipSuccess) { *p = nullptr; return qh_fail(QH_ENOMEM, "hipHostMalloc of %zu bytes failed: %s", want, hipGetErrorString(e)); }
  *cap = want;
  return QH_OK;
}

extern "C" void qh_corrector_destroy(qh_corrector* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  void* dptrs[] = {c->own_table, c->d_cbits, c->d_P, c->d_ptab, c->d_chunk, c->d_recs, c->d_hdr, c->d_tiles, c->d_nlplane, c->d_obuf, c->d_logs,
                   c->d_res, c->d_bsums, c->d_fa, c->d_log, c->d_meta, c->d_tot};
  for (void* p : dptrs) if (p) (void)hipFree(p);
  void* hptrs[] = {c->h_fa[0], c->h_fa[1], c->h_log[0], c->h_log[1], c->h_meta, c->h_tot};
  for (void* p : hptrs) if (p) (void)hipHostFree(p);
  for (hipEvent_t e : c->ev) if (e) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

static int corrector_init(qh_corrector* c) {
  QH_HIP(hipSetDevice(c->device));
  QH_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  QH_HIP(hipMalloc((void**)&c->d_meta, sizeof(qh_meta)));
  QH_HIP(hipHostMalloc((void**)&c->h_meta, sizeof(qh_meta), hipHostMallocDefault));
  QH_HIP(hipMalloc((void**)&c->d_tot, sizeof(qh_ec_totals)));
  QH_HIP(hipHostMalloc((void**)&c->h_tot, sizeof(qh_ec_totals), hipHostMallocDefault));
  for (hipEvent_t& e : c->ev) QH_HIP(hipEventCreate(&e));
  // Poisson decision table (error_correct_reads.cc:445-453): the test is only reached with
  // min_count < counts[ori] < cutoff and the sum of four counts <= 4 max_val
  const qh_correct_params& p = c->params;
  const uint64_t max_val = c->db->g.max_val;
  const uint64_t stride = max_val <= (1ull << 24) ? 4 * max_val + 1 : 0;
  const uint64_t rows = p.cutoff > 0 ? (uint64_t)p.cutoff : 0;
  if (stride && rows && rows * stride <= (1ull << 24)) {
    std::vector<uint32_t> tab((size_t)((rows * stride + 31) / 32), 0u);
    for (uint64_t co = 0; co < rows; ++co)
      for (uint64_t sum = 0; sum < stride; ++sum)
        if (qh_poisson_term((double)sum * p.collision_prob, (unsigned)co) < p.poisson_threshold) {
          const uint64_t bit = co * stride + sum;
          tab[(size_t)(bit >> 5)] |= 1u << (bit & 31);
        }
    QH_HIP(hipMalloc((void**)&c->d_ptab, tab.size() * 4));
    QH_HIP(hipMemcpy(c->d_ptab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    c->P.ptab_stride = (int)stride;
    c->P.ptab = c->d_ptab;
  }
  // The corrector's unit of work is a walk from a k-mer's home slot to the first 64-byte line with an empty slot, and
  // its length follows the load of the table: against the bench database at load 0.54 the kernel takes 7.2 ms, against
  // the same content in twice the slots 5.9 ms. The database as built stays as it is (file, query handle); the
  // corrector works on a private copy re-hashed into the smallest table with a load of at most QH_EC_TABLE_LOAD
  // (default 0.3; 0: never) when that copy takes less than a quarter of the free HBM. Same keys, same values: every
  // lookup returns what it returned before.
  {
    const char* ev = getenv("QH_EC_TABLE_LOAD");
    const double target = ev ? atof(ev) : 0.3;
    const qh_geom og = c->P.g;
    if (target > 0 && og.shard_bits == 0 && !og.wide) {
      QH_HIP(hipMemsetAsync(c->d_meta, 0, sizeof(qh_meta), c->stream));
      qh_launch_stats(c->P.table, og, c->d_meta, c->stream);
      QH_HIP(hipMemcpyAsync(c->h_meta, c->d_meta, sizeof(qh_meta), hipMemcpyDeviceToHost, c->stream));
      QH_HIP(hipStreamSynchronize(c->stream));
      // every entry fills a slot, whatever its quality (low-quality-only mers are often the majority of a database
      // built from raw reads): the copy is sized by the occupied slots
      const double distinct = (double)c->h_meta->stat_occupied;
      int lsize = og.lsize;
      while (lsize < 40 && distinct > target * (double)(1ull << lsize)) ++lsize;
      size_t free_b = 0, total_b = 0;
      QH_HIP(hipMemGetInfo(&free_b, &total_b));
      qh_geom ng;
      std::string err;
      if (lsize > og.lsize && ((size_t)8 << lsize) < free_b / 4 &&
          qh_make_geom(og.k, og.bits, 1ull << lsize, og.min_qual, (int)og.max_reprobe, 0, 0, -1, ng, err)) {
        uint64_t* nt = nullptr;
        if (hipMalloc((void**)&nt, (size_t)8 << ng.lsize) == hipSuccess) {
          QH_HIP(hipMemsetAsync(nt, 0, (size_t)8 << ng.lsize, c->stream));
          QH_HIP(hipMemsetAsync(c->d_meta, 0, sizeof(qh_meta), c->stream));
          qh_launch_rehash(c->P.table, og, nt, ng, c->d_meta, c->stream);
          QH_HIP(hipMemcpyAsync(c->h_meta, c->d_meta, sizeof(qh_meta), hipMemcpyDeviceToHost, c->stream));
          QH_HIP(hipStreamSynchronize(c->stream));
          if (c->h_meta->rehash_fail) { (void)hipFree(nt); }  // a key found no place: stay on the database's own table
          else { c->own_table = nt; c->P.g = ng; c->P.table = nt; }
        } else {
          (void)hipGetLastError();
        }
      }
    }
  }
  if (c->P.ctable && !getenv("QH_EC_NO_CONTAMINANT_BITS")) {
    // 2^22 bits (512 KiB, at home in the L2 cache): a set of a few thousand k-mers passes one base in a thousand on.
    // (Taking the index from the core hash the sibling walk computes anyway, so that a base that passes is not
    // hashed at all, was measured too: no gain over testing the bit after the hash.)
    int b = 22;
    if (b > 2 * c->P.cg.k) b = 2 * c->P.cg.k;
    const size_t words = b > 5 ? (size_t)1 << (b - 5) : 1;
    if (hipMalloc((void**)&c->d_cbits, words * 4) == hipSuccess) {
      QH_HIP(hipMemsetAsync(c->d_cbits, 0, words * 4, c->stream));
      const uint32_t mask = (uint32_t)(((uint64_t)1 << b) - 1);
      hipLaunchKernelGGL(k_ec_contaminant_bits, dim3(256), dim3(256), 0, c->stream, c->P.ctable, c->P.cg,
                         qh_table_slots(c->P.cg), c->d_cbits, mask);
      QH_HIP(hipGetLastError());
      QH_HIP(hipStreamSynchronize(c->stream));
      c->P.cbits = c->d_cbits;
      c->P.cbits_mask = mask;
    } else {
      (void)hipGetLastError();
    }
  }
  const qh_geom& g = c->P.g;
  const qh_geom& cg = c->P.cg;
  const bool val32 = g.vbits <= 32 && (!c->P.ctable || cg.vbits <= 32);
  const bool tag32 = g.rbits + g.pbits <= 32 && (!c->P.ctable || cg.rbits + cg.pbits <= 32);
  c->word_mode = (g.wide || (c->P.ctable && cg.wide)) ? 3 : val32 ? (tag32 ? 0 : 1) : 2;
  c->siblings = g.k >= 4 && g.wbits >= 3 && g.lsize + g.shard_bits <= 2 * (g.k - 2);
  QH_HIP(hipMalloc((void**)&c->d_P, sizeof(qh_ec_dev)));
  QH_HIP(hipMemcpy(c->d_P, &c->P, sizeof(qh_ec_dev), hipMemcpyHostToDevice));
  return QH_OK;
}

extern "C" int qh_corrector_create(qh_corrector** out, qh_query* db, qh_query* contaminant, const qh_correct_params* p) {
  if (!out || !db || !p) return qh_fail(QH_EBADARG, "null argument");
  *out = nullptr;
  if (contaminant) {
    if (contaminant->g.k != db->g.k)  // error_correct_reads.cc:703-705
      return qh_fail(QH_EBADARG, "Contaminant mer length (%d) different than correction mer length (%d)", contaminant->g.k, db->g.k);
    if (contaminant->device != db->device) return qh_fail(QH_EBADARG, "database and contaminant set live on different devices");
  }
  if (p->skip < 0 || p->good < 0 || p->anchor < 0 || p->min_count < 0 || p->cutoff < 0 || p->window < 0 || p->error < 0)
    return qh_fail(QH_EBADARG, "negative corrector parameter");
  if (p->window >= (1 << 23)) return qh_fail(QH_EBADARG, "window too large");
  qh_corrector* c = new qh_corrector();
  c->device = db->device; c->db = db; c->contaminant = contaminant; c->params = *p;
  qh_ec_dev& P = c->P;
  memset(&P, 0, sizeof P);
  P.g = db->g; P.table = db->table;
  if (contaminant) { P.cg = contaminant->g; P.ctable = contaminant->table; }
  P.skip = p->skip; P.good = p->good; P.anchor = p->anchor; P.min_count = p->min_count; P.cutoff = p->cutoff;
  P.window = p->window ? p->window : db->g.k;   // error_correct_reads.cc:206
  P.error = p->error ? p->error : db->g.k / 2;  // :207
  P.qual_cutoff = p->qual_cutoff; P.trim_contaminant = p->trim_contaminant; P.do_homo_trim = p->do_homo_trim;
  P.homo_trim = p->homo_trim; P.no_discard = p->no_discard;
  P.collision_prob = p->collision_prob; P.poisson_threshold = p->poisson_threshold;
  int rc = corrector_init(c);
  if (rc) { qh_corrector_destroy(c); return rc; }
  *out = c;
  return QH_OK;
}

static const size_t kMaxEcChunk = 0x7FFFF000ull;  // 31-bit offsets: positions are ints in the kernel

static void launch_format(qh_corrector* c) {
  const uint32_t nblocks = (c->cur_rec_cap + EC_FMT_BLOCK - 1) / EC_FMT_BLOCK;
  hipLaunchKernelGGL(k_ec_format, dim3(nblocks), dim3(EC_FMT_BLOCK), 0, c->stream, c->cur_buf, c->d_recs, c->cur_hdr,
                     c->d_meta, c->d_obuf, c->d_logs, c->d_res, c->d_bsums, c->d_tot, c->d_fa, c->d_log);
}

// record table (and nrec in d_meta) ready on the stream: correct + sizes + scan + format
static int enqueue_correct(qh_corrector* c, const uint8_t* dev_buf, size_t len, uint32_t rec_cap, const uint2* d_hdr,
                           size_t log_words) {
  int rc;
  if ((rc = ensure_dev((void**)&c->d_obuf, &c->obuf_cap, len + 16, "corrected bases"))) return rc;
  if ((rc = ensure_dev((void**)&c->d_logs, &c->logs_cap, log_words * 4, "error logs"))) return rc;
  if ((rc = ensure_dev((void**)&c->d_res, &c->res_cap, (size_t)rec_cap * EC_RES_WORDS * 4 + 16, "read results"))) return rc;
  const uint32_t nblocks = (rec_cap + EC_FMT_BLOCK - 1) / EC_FMT_BLOCK;
  if ((rc = ensure_dev((void**)&c->d_bsums, &c->bsums_cap, ((size_t)nblocks + 1) * 16, "block sums"))) return rc;
  // first guess of the output sizes; qh_corrector_fetch grows them and writes the text again when it was short
  if ((rc = ensure_dev((void**)&c->d_fa, &c->fa_cap, len + 4096, "corrected reads text"))) return rc;
  if ((rc = ensure_dev((void**)&c->d_log, &c->log_cap, len / 8 + 65536, "skip log text"))) return rc;
  qh_ec_totals z;
  memset(&z, 0, sizeof z);
  z.fa_cap = c->fa_cap; z.log_cap = c->log_cap;
  *c->h_tot = z;
  QH_HIP(hipMemcpyAsync(c->d_tot, c->h_tot, sizeof z, hipMemcpyHostToDevice, c->stream));
  c->cur_buf = dev_buf; c->cur_hdr = d_hdr; c->cur_rec_cap = rec_cap;
  QH_HIP(hipEventRecord(c->ev[1], c->stream));
  unsigned grid = (rec_cap + EC_BLOCK - 1) / EC_BLOCK;
  if (grid > 256u * 16u) grid = 256u * 16u;
  if (grid < 1) grid = 1;
  auto launch = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(EC_BLOCK), 0, c->stream, dev_buf, len, c->d_recs, d_hdr, c->d_meta, c->d_P,
                       c->P.table, c->P.ctable, c->P.ptab, c->d_obuf, c->d_logs, c->d_res, c->d_tot);
  };
  switch (c->word_mode * 2 + (c->siblings ? 1 : 0)) {
    case 0: launch(k_correct_reads<0, false>); break;
    case 1: launch(k_correct_reads<0, true>); break;
    case 2: launch(k_correct_reads<1, false>); break;
    case 3: launch(k_correct_reads<1, true>); break;
    case 4: launch(k_correct_reads<2, false>); break;
    case 5: launch(k_correct_reads<2, true>); break;
    case 6: launch(k_correct_reads<3, false>); break;
    default: launch(k_correct_reads<3, true>); break;
  }
  QH_HIP(hipEventRecord(c->ev[2], c->stream));
  hipLaunchKernelGGL(k_ec_block_sums, dim3(nblocks ? nblocks : 1), dim3(EC_FMT_BLOCK), 0, c->stream, c->d_res, c->d_meta,
                     c->d_bsums);
  hipLaunchKernelGGL(k_ec_scan_blocks, dim3(1), dim3(1024), 0, c->stream, c->d_bsums, nblocks ? nblocks : 1, c->d_meta, c->d_tot);
  launch_format(c);
  QH_HIP(hipEventRecord(c->ev[3], c->stream));
  QH_HIP(hipGetLastError());
  c->pending = true;
  return QH_OK;
}

// An error-log entry keeps the position of a substitution or truncation in 24 bits (EC_POS): a read of 2^24 bases or
// more would corrupt its log. Such a read rejects the chunk (QH_EFORMAT) before anything is corrected.
#define EC_MAX_READ_LEN (1u << 24)
__global__ __launch_bounds__(256) void k_ec_check_lengths(const uint4* __restrict__ recs, qh_meta* __restrict__ meta) {
  if (meta->format_err) return;
  const uint32_t nrec = meta->nrec;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nrec; r += gridDim.x * blockDim.x)
    if (recs[r].y >= EC_MAX_READ_LEN) {
      atomicOr(&meta->format_err, QH_FMT_LINE_TOO_LONG);
      atomicMin(&meta->first_bad, r);
    }
}

static int correct_device_on(qh_corrector* c, const void* dev_buf, size_t len) {
  if (len == 0) return qh_fail(QH_EBADARG, "empty chunk");
  if (!dev_buf || ((uintptr_t)dev_buf & 15)) return qh_fail(QH_EBADARG, "device chunk must be non-null and 16-byte aligned");
  if (len > kMaxEcChunk) return qh_fail(QH_EBADARG, "chunk larger than %zu bytes", kMaxEcChunk);
  QH_HIP(hipSetDevice(c->device));
  int rc;
  const size_t ntiles = (len + QH_TILE_BYTES - 1) / QH_TILE_BYTES;
  const size_t rec_cap = len / 16 + 16;
  if ((rc = ensure_dev((void**)&c->d_tiles, &c->tile_cap, (ntiles + 1 + QH_SCAN_SCRATCH) * 4, "tile counts"))) return rc;
  if ((rc = ensure_dev((void**)&c->d_recs, &c->rec_cap, rec_cap * 16, "record table"))) return rc;
  if ((rc = ensure_dev((void**)&c->d_nlplane, &c->nlplane_cap, (len / 16 + 2) * 8 + 8192, "newline lists"))) return rc;
  QH_HIP(hipEventRecord(c->ev[0], c->stream));
  QH_HIP(hipMemsetAsync(c->d_meta, 0, sizeof(qh_meta), c->stream));
  qh_launch_index_fastq((const uint8_t*)dev_buf, len, c->d_tiles, c->d_nlplane, c->d_recs, (uint32_t)rec_cap, c->d_meta,
                        nullptr, 0, c->stream);
  hipLaunchKernelGGL(k_ec_check_lengths, dim3(256), dim3(256), 0, c->stream, c->d_recs, c->d_meta);
  // log words: seq_off + 4 r + len + 4 <= len_chunk + 4 nrec + 8, nrec <= len / 6 for non-empty 4-line records
  return enqueue_correct(c, (const uint8_t*)dev_buf, len, (uint32_t)rec_cap, nullptr, len + 4 * (len / 6 + 2) + 64);
}

extern "C" int qh_corrector_correct_device(qh_corrector* c, const void* dev_buf, size_t len, void* hip_stream) {
  if (!c) return qh_fail(QH_EBADARG, "null handle");
  if (hip_stream) {  // order this chunk after the caller's stream
    QH_HIP(hipSetDevice(c->device));
    hipEvent_t e;
    QH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipError_t er = hipEventRecord(e, (hipStream_t)hip_stream);
    if (er == hipSuccess) er = hipStreamWaitEvent(c->stream, e, 0);
    (void)hipEventDestroy(e);
    if (er != hipSuccess) return qh_fail(QH_EHIP, "ordering after the caller's stream failed: %s", hipGetErrorString(er));
  }
  return correct_device_on(c, dev_buf, len);
}

static const char* format_error_text(uint32_t err) {
  if (err & QH_FMT_NO_AT) return "Invalid fastq file: header missing";
  if (err & QH_FMT_LINES_MOD4) return "chunk is not made of whole 4-line records";
  if (err & QH_FMT_NO_PLUS) return "Invalid fastq file: '+' line missing (multi-line record?)";
  if (err & QH_FMT_LEN_MISMATCH) return "Invalid fastq file: wrong number of quals";
  if (err & QH_FMT_EMPTY_TAIL) return "empty read";
  if (err & QH_FMT_LINE_TOO_LONG) return "a read of 2^24 bases or more (the error log keeps positions in 24 bits)";
  return "too many records";
}

extern "C" int qh_corrector_fetch(qh_corrector* c, const char** fa, size_t* fa_len, const char** log, size_t* log_len,
                                  qh_correct_stats* stats) {
  if (!c) return qh_fail(QH_EBADARG, "null handle");
  if (fa) *fa = nullptr;
  if (fa_len) *fa_len = 0;
  if (log) *log = nullptr;
  if (log_len) *log_len = 0;
  if (stats) memset(stats, 0, sizeof *stats);
  if (!c->pending) return QH_OK;
  c->pending = false;
  QH_HIP(hipSetDevice(c->device));
  QH_HIP(hipMemcpyAsync(c->h_meta, c->d_meta, sizeof(qh_meta), hipMemcpyDeviceToHost, c->stream));
  QH_HIP(hipMemcpyAsync(c->h_tot, c->d_tot, sizeof(qh_ec_totals), hipMemcpyDeviceToHost, c->stream));
  QH_HIP(hipStreamSynchronize(c->stream));
  if (c->h_meta->format_err)
    return qh_fail(QH_EFORMAT, "%s (record %u of the chunk)", format_error_text(c->h_meta->format_err), c->h_meta->first_bad);
  if (c->h_tot->overflow) {  // the first guess was short: grow and write the text again
    int rc;
    if ((rc = ensure_dev((void**)&c->d_fa, &c->fa_cap, c->h_tot->fa_bytes + 16, "corrected reads text"))) return rc;
    if ((rc = ensure_dev((void**)&c->d_log, &c->log_cap, c->h_tot->log_bytes + 16, "skip log text"))) return rc;
    c->h_tot->fa_cap = c->fa_cap; c->h_tot->log_cap = c->log_cap; c->h_tot->overflow = 0;
    QH_HIP(hipMemcpyAsync(c->d_tot, c->h_tot, sizeof(qh_ec_totals), hipMemcpyHostToDevice, c->stream));
    launch_format(c);
    QH_HIP(hipGetLastError());
  }
  const size_t nfa = (size_t)c->h_tot->fa_bytes, nlog = (size_t)c->h_tot->log_bytes;
  int rc;
  const int o = c->h_out;
  c->h_out ^= 1;
  if ((rc = ensure_pinned(&c->h_fa[o], &c->h_fa_cap[o], nfa + 1))) return rc;
  if ((rc = ensure_pinned(&c->h_log[o], &c->h_log_cap[o], nlog + 1))) return rc;
  if (nfa) QH_HIP(hipMemcpyAsync(c->h_fa[o], c->d_fa, nfa, hipMemcpyDeviceToHost, c->stream));
  if (nlog) QH_HIP(hipMemcpyAsync(c->h_log[o], c->d_log, nlog, hipMemcpyDeviceToHost, c->stream));
  QH_HIP(hipStreamSynchronize(c->stream));
  c->h_fa[o][nfa] = 0; c->h_log[o][nlog] = 0;
  if (fa) *fa = c->h_fa[o];
  if (fa_len) *fa_len = nfa;
  if (log) *log = c->h_log[o];
  if (log_len) *log_len = nlog;
  if (stats) {
    stats->reads = c->h_tot->reads; stats->corrected = c->h_tot->corrected; stats->skipped = c->h_tot->skipped;
    stats->substitutions = c->h_tot->substitutions; stats->truncations = c->h_tot->truncations;
    stats->bases_in = c->h_tot->bases_in; stats->bases_out = c->h_tot->bases_out;
  }
  return QH_OK;
}

extern "C" int qh_corrector_kernel_time(qh_corrector* c, double* correct_ms, double* chunk_ms) {
  if (!c) return qh_fail(QH_EBADARG, "null handle");
  QH_HIP(hipSetDevice(c->device));
  QH_HIP(hipEventSynchronize(c->ev[3]));
  float a = 0, b = 0;
  QH_HIP(hipEventElapsedTime(&a, c->ev[1], c->ev[2]));
  QH_HIP(hipEventElapsedTime(&b, c->ev[0], c->ev[3]));
  if (correct_ms) *correct_ms = a;
  if (chunk_ms) *chunk_ms = b;
  return QH_OK;
}

extern "C" int qh_corrector_correct_fastq(qh_corrector* c, const char* buf, size_t len, int final_chunk, size_t* consumed) {
  if (!c) return qh_fail(QH_EBADARG, "null handle");
  if (consumed) *consumed = 0;
  if (len == 0) return QH_OK;
  if (!buf) return qh_fail(QH_EBADARG, "null chunk");
  if (len > kMaxEcChunk / 2) return qh_fail(QH_EBADARG, "chunk larger than %zu bytes", kMaxEcChunk / 2);
  QH_HIP(hipSetDevice(c->device));
  QH_HIP(hipStreamSynchronize(c->stream));  // staging buffers are reused
  c->pending = false;
  int rc;
  // fast path: the leading whole 4-line groups, indexed and checked on the GPU
  const size_t used = qh_fastq_4line_prefix(buf, len, final_chunk);
  if (used) {
    if ((rc = ensure_dev((void**)&c->d_chunk, &c->chunk_cap, used + 32, "chunk staging"))) return rc;
    QH_HIP(hipMemcpyAsync(c->d_chunk, buf, used, hipMemcpyHostToDevice, c->stream));
    if ((rc = correct_device_on(c, c->d_chunk, used))) return rc;
    QH_HIP(hipMemcpyAsync(c->h_meta, c->d_meta, sizeof(qh_meta), hipMemcpyDeviceToHost, c->stream));
    QH_HIP(hipStreamSynchronize(c->stream));
    if (!c->h_meta->format_err) {
      if (consumed) *consumed = used;
      return QH_OK;
    }
    c->pending = false;  // rejected: nothing was corrected (the kernels return on format_err)
  }
  // sequential host parser with Jellyfish's rules (multi-line records)
  std::vector<qh_record> recs;
  std::vector<qh_hdr> hdrs;
  std::string scratch;
  size_t eaten = 0;
  if ((rc = qh_parse_fastq_host(buf, len, final_chunk != 0, recs, scratch, &eaten, &hdrs))) return rc;
  if (consumed) *consumed = eaten;
  if (recs.empty()) return QH_OK;
  const size_t eo = (eaten + 15) & ~(size_t)15;
  if ((rc = ensure_dev((void**)&c->d_chunk, &c->chunk_cap, eo + scratch.size() + 32, "chunk staging"))) return rc;
  size_t log_words = 0;
  for (size_t i = 0; i < recs.size(); ++i) {
    if (recs[i].reserved & 1u) recs[i].seq_off += (uint32_t)eo;
    if (recs[i].reserved & 2u) recs[i].qual_off += (uint32_t)eo;
    if (recs[i].len >= EC_MAX_READ_LEN)
      return qh_fail(QH_EFORMAT, "read %zu of the chunk has %u bases: the corrector takes reads shorter than 2^24 bases "
                                 "(its error log keeps positions in 24 bits)", i, recs[i].len);
    if (log_words + recs[i].len + 4 > 0xFFFFFFF0ull) return qh_fail(QH_EBADARG, "chunk too large for the error logs");
    recs[i].reserved = (uint32_t)log_words;  // becomes .w of the device record: the read's log offset
    log_words += (size_t)recs[i].len + 4;
  }
  if ((rc = ensure_dev((void**)&c->d_recs, &c->rec_cap, (recs.size() + 1) * 16, "record table"))) return rc;
  if ((rc = ensure_dev((void**)&c->d_hdr, &c->hdr_cap, (recs.size() + 1) * 8, "header table"))) return rc;
  QH_HIP(hipMemcpyAsync(c->d_chunk, buf, eaten, hipMemcpyHostToDevice, c->stream));
  if (!scratch.empty()) QH_HIP(hipMemcpyAsync(c->d_chunk + eo, scratch.data(), scratch.size(), hipMemcpyHostToDevice, c->stream));
  static_assert(sizeof(qh_record) == 16 && sizeof(qh_hdr) == 8, "host records must match the device tables");
  QH_HIP(hipMemcpyAsync(c->d_recs, recs.data(), recs.size() * 16, hipMemcpyHostToDevice, c->stream));
  QH_HIP(hipMemcpyAsync(c->d_hdr, hdrs.data(), hdrs.size() * 8, hipMemcpyHostToDevice, c->stream));
  memset(c->h_meta, 0, sizeof(qh_meta));
  c->h_meta->nrec = (uint32_t)recs.size();
  QH_HIP(hipMemcpyAsync(c->d_meta, c->h_meta, sizeof(qh_meta), hipMemcpyHostToDevice, c->stream));
  QH_HIP(hipEventRecord(c->ev[0], c->stream));
  rc = enqueue_correct(c, c->d_chunk, eo + scratch.size() + 16, (uint32_t)recs.size(), c->d_hdr, log_words + 16);
  QH_HIP(hipStreamSynchronize(c->stream));  // recs / hdrs / scratch are locals
  return rc;
}
