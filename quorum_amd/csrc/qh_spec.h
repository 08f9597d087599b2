// qh_spec.h -- table geometry, k-mer codec and position hash shared by the host
// code and the gfx950 kernels of libquorum_hip.
//
// Upstream semantics this replaces (paths relative to the Quorum tree):
//   jellyfish::mer_dna codec used at src/create_database.cc:73-86
//   jellyfish::large_hash::array geometry used at src/mer_database.hpp:78-81
//   value field "count<<1 | hq" and max_val    src/mer_database.hpp:83,102-111
//
// Slot design (one 64-bit word per slot, claimed AND updated by one 64-bit CAS):
//
//   63 ............................................................. 0
//   [ remainder : rbits ][ probe+1 : pbits ][ value = count<<1|hq : vbits ]
//
// pk  = qh_pkey(canonical k-mer)           the key's two bit planes side by side:
//                                          low bits of the k codes | high bits << k
//                                          (a fixed bit permutation of the 2-bit packing;
//                                          it is what the wave-parallel extractor produces)
// h   = qh_mix(pk)                         injection of the 2k-bit keys into 2k+1-bit integers whose low
//                                          bits depend only on the canonical CORE (bases 1..k-2)
// owner shard = h & (2^shard_bits - 1)     (0 when the table is not sharded; it does
//                                           not depend on lsize, so a shard can double)
// local home  = (h >> shard_bits) & (2^lsize - 1)
// remainder   = h >> gsize                 gsize = lsize + shard_bits, rbits = max(0, 2k + 1 - gsize)
// probing     = linear inside the 2^wbits-slot window that contains the home
//               slot: idx = (home & ~wmask) | ((home + i) & wmask), i <= max_reprobe
// A slot is empty iff the whole word is 0 (an occupied slot has probe+1 >= 1).
// Wide form (qh_geom::wide): vbits = 0, the slot is [ remainder | probe+1 ] and the value of slot i is wvals[i].
// The key of an occupied slot is recovered as
//   qh_unmix( remainder << gsize | ((window base | (idx - i) & wmask)) << shard_bits | shard ).
// A table that holds ALL shards back to back (all_shards = 1: the serialised file,
// a query handle) indexes slot (shard << lsize) | local home.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define QH_HD __host__ __device__ __forceinline__
#else
#define QH_HD inline
#endif

#define QH_WINDOW_BITS 13          // 8192 slots = 64 KiB: two windows (two workgroups) fit a CU's LDS
#define QH_MIX_C1 0x9E3779B1u
#define QH_MIX_C2 0x85EBCA77u
#define QH_MIX_C3 0xC2B2AE3Du
// XORed into the core's hash: with it the all-ones 63-bit pattern is the hash of NO 31-mer (its preimage
// is a core larger than its reverse complement, which is never the one hashed), so the all-ones item
// (hash << 1 | 1 = ~0) stays free as the "no item" mark of the partition kernels at k = 31.
// tests/test_dist_cpu.py checks this.
#define QH_MIX_Z 5u

struct qh_geom {
  int32_t k;            // mer length, 1..31
  int32_t bits;         // count bits (-b); value field is bits+1 wide
  int32_t vbits;        // width of the value field INSIDE the slot: bits + 1, or 0 in the wide form
  int32_t pbits;        // width of the probe+1 field
  int32_t rbits;        // width of the remainder field
  int32_t lsize;        // log2(slots in THIS shard)
  int32_t shard_bits;   // log2(number of shards)
  int32_t shard_id;     // which shard this table is
  int32_t wbits;        // log2(probe window) = min(lsize, QH_WINDOW_BITS)
  int32_t max_reprobe;  // probe offsets 0..max_reprobe are tried
  int32_t min_qual;     // threshold, compared as signed char
  int32_t all_shards;   // 1: this table is the concatenation of all 2^shard_bits shards
  uint64_t kmask;       // 2^(2k) - 1
  uint64_t hmask;       // 2^(2k+1) - 1: the hash of a key is one bit wider than the key
  uint64_t vmask;       // 2^vbits - 1
  uint64_t max_val;     // 2^bits - 1
  // Wide form (remainder + probe + value fields would exceed 64 bits: large -b, or long mers in small tables): the
  // slot keeps [remainder | probe+1] only and the bits+1-wide value lives in a side array of one 64-bit word per slot,
  // like upstream's separate val_array (src/mer_database.hpp:78-83). `wvals` is that array's device address: it rides
  // in the geometry so that every kernel that is handed a table can reach the values that belong to it.
  int32_t wide;
  int32_t reserved_;
  uint64_t* wvals;
};

QH_HD uint64_t qh_low_mask(int n) { return n >= 64 ? ~0ULL : ((1ULL << n) - 1); }

// Position hash ("feistel3-core/2"). The key is the k-mer AS GIVEN (hash_with_quality::add does not
// canonicalise, src/mer_database.hpp:94), seen as plane halves L = low code bits, R = high code bits.
// Its CORE = bases 1 .. k-2, taken in whichever orientation (as is, or reverse complemented) is the
// smaller 2(k-2)-bit number, goes through a 3-round Feistel network (round function = top k-2 bits of a
// 32-bit odd multiply; 64-bit multiplies are quarter rate on CDNA) and supplies the LOW bits of the hash:
// shard, window and home slot. The four end-base bits and the orientation bit sit above it:
//
//   h = [ flip : 1 ][ ends : 4 ][ Feistel(canonical core) : 2(k-2) ]            (2k + 1 bits)
//
// A Feistel network is a bijection, so h determines the key (qh_unmix) and the stored remainder + home
// position reproduce it. Why the core: the corrector asks for the four k-mers that differ in their first
// or last base (database_query::get_best_alternatives, src/mer_database.hpp:302-329) at every base of
// every read; after canonicalisation those siblings are spread over both orientations, but all of them
// share the canonical core, hence the home slot, hence (linear probing) one or two 64-byte lines of
// HBM instead of four random ones. For k < 4 there is no core: the whole key goes through the network.
QH_HD uint32_t qh_brev32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __brev(x);
#else
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
  x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
  x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
  x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
  return (x >> 16) | (x << 16);
#endif
}

// planes of the reverse complement of an n-base sequence given as planes (base i at bit i), 1 <= n <= 31
QH_HD void qh_rc_planes(uint32_t L, uint32_t R, int n, uint32_t* rL, uint32_t* rR) {
  *rL = (~qh_brev32(L)) >> (32 - n);
  *rR = (~qh_brev32(R)) >> (32 - n);
}

// (aL, aR) > (bL, bR) as 2-bit packed integers: decided at the most significant base that differs
QH_HD bool qh_planes_greater(uint32_t aL, uint32_t aR, uint32_t bL, uint32_t bR) {
  const uint32_t d1 = aR ^ bR, d = (aL ^ bL) | d1;
  if (d == 0) return false;
  const uint32_t bit = 0x80000000u >> __builtin_clz(d);
  return ((d1 & bit) ? (aR & bit) : (aL & bit)) != 0;
}

QH_HD uint64_t qh_feistel3(uint32_t L, uint32_t R, int n) {
  const int sh = 32 - n;
  R ^= (L * QH_MIX_C1) >> sh;
  L ^= (R * QH_MIX_C2) >> sh;
  R ^= (L * QH_MIX_C3) >> sh;
  return (uint64_t)R | ((uint64_t)L << n);
}

// hash of the key with planes (L, R) whose reverse complement has planes (oL, oR)
QH_HD uint64_t qh_mix_pair(uint32_t L, uint32_t R, uint32_t oL, uint32_t oR, int k) {
  if (k < 4) return qh_feistel3(L, R, k);
  const int kc = k - 2;
  const uint32_t cm = (1u << kc) - 1u;
  uint32_t cL = (L >> 1) & cm, cR = (R >> 1) & cm;
  const uint32_t dL = (oL >> 1) & cm, dR = (oR >> 1) & cm;
  const bool flip = qh_planes_greater(cL, cR, dL, dR);
  if (flip) { cL = dL; cR = dR; }
  const uint32_t e = (L & 1u) | ((R & 1u) << 1) | ((L >> (k - 1)) << 2) | ((R >> (k - 1)) << 3);
  return (qh_feistel3(cL, cR, kc) ^ QH_MIX_Z) | ((uint64_t)e << (2 * kc)) | ((uint64_t)flip << (2 * kc + 4));
}

QH_HD uint64_t qh_mix2(uint32_t L, uint32_t R, int k) {
  uint32_t oL, oR;
  qh_rc_planes(L, R, k, &oL, &oR);
  return qh_mix_pair(L, R, oL, oR, k);
}

QH_HD uint64_t qh_mix(uint64_t x, int k) {
  const uint32_t km = (uint32_t)((1ull << k) - 1);
  return qh_mix2((uint32_t)x & km, (uint32_t)(x >> k) & km, k);
}

QH_HD uint64_t qh_unfeistel3(uint64_t x, int n) {
  const uint32_t nm = (uint32_t)((1ull << n) - 1);
  const int sh = 32 - n;
  uint32_t R = (uint32_t)x & nm, L = (uint32_t)(x >> n) & nm;
  R ^= (L * QH_MIX_C3) >> sh;
  L ^= (R * QH_MIX_C2) >> sh;
  R ^= (L * QH_MIX_C1) >> sh;
  return (uint64_t)L | ((uint64_t)R << n);
}

// hash -> plane key (low code bits | high code bits << k)
QH_HD uint64_t qh_unmix(uint64_t h, int k) {
  if (k < 4) return qh_unfeistel3(h, k);
  const int kc = k - 2;
  const uint32_t cm = (1u << kc) - 1u;
  const uint64_t c = qh_unfeistel3((h & qh_low_mask(2 * kc)) ^ QH_MIX_Z, kc);
  uint32_t cL = (uint32_t)c & cm, cR = (uint32_t)(c >> kc) & cm;
  const uint32_t e = (uint32_t)(h >> (2 * kc)) & 15u;
  if ((h >> (2 * kc + 4)) & 1u) qh_rc_planes(cL, cR, kc, &cL, &cR);
  const uint32_t L = (cL << 1) | (e & 1u) | (((e >> 2) & 1u) << (k - 1));
  const uint32_t R = (cR << 1) | ((e >> 1) & 1u) | (((e >> 3) & 1u) << (k - 1));
  return (uint64_t)L | ((uint64_t)R << k);
}

// reverse complement of a k-mer in the low 2k bits (A=0,C=1,G=2,T=3: comp = 3-c = ~c)
QH_HD uint64_t qh_revcomp(uint64_t m, int k) {
  uint64_t x = ~m;
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  x = ((x >> 8) & 0x00FF00FF00FF00FFULL) | ((x & 0x00FF00FF00FF00FFULL) << 8);
  x = ((x >> 16) & 0x0000FFFF0000FFFFULL) | ((x & 0x0000FFFF0000FFFFULL) << 16);
  x = (x >> 32) | (x << 32);
  return x >> (64 - 2 * k);
}

// even bits of x gathered into the low 32 bits
QH_HD uint32_t qh_compress_even(uint64_t x) {
  x &= 0x5555555555555555ULL;
  x = (x | (x >> 1)) & 0x3333333333333333ULL;
  x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0FULL;
  x = (x | (x >> 4)) & 0x00FF00FF00FF00FFULL;
  x = (x | (x >> 8)) & 0x0000FFFF0000FFFFULL;
  x = (x | (x >> 16)) & 0x00000000FFFFFFFFULL;
  return (uint32_t)x;
}

// 32 bits spread to the even bit positions of a 64-bit word
QH_HD uint64_t qh_spread_even(uint32_t v) {
  uint64_t x = v;
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL;
  x = (x | (x << 8)) & 0x00FF00FF00FF00FFULL;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0FULL;
  x = (x | (x << 2)) & 0x3333333333333333ULL;
  x = (x | (x << 1)) & 0x5555555555555555ULL;
  return x;
}

// 2-bit packed k-mer <-> plane key (low code bits | high code bits << k)
QH_HD uint64_t qh_pkey(uint64_t key, int k) {
  return (uint64_t)qh_compress_even(key) | ((uint64_t)qh_compress_even(key >> 1) << k);
}

QH_HD uint64_t qh_key_from_pkey(uint64_t pk, int k) {
  const uint32_t km = (uint32_t)((1ULL << k) - 1);
  return qh_spread_even((uint32_t)pk & km) | (qh_spread_even((uint32_t)(pk >> k) & km) << 1);
}

QH_HD uint64_t qh_canonical(uint64_t m, int k) {
  const uint64_t r = qh_revcomp(m, k);
  return m < r ? m : r;
}

// mer_dna::code: A/a C/c G/g T/t -> 0..3, everything else negative
QH_HD int qh_code(unsigned c) {
  const unsigned u = c & 0xDFu;
  return u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : u == 'T' ? 3 : -1;
}

struct qh_pos {
  uint64_t home;   // home slot inside this shard
  uint64_t tag0;   // remainder << pbits (probe field still 0), NOT shifted by vbits
  uint32_t shard;  // owner shard of the key
};

QH_HD uint64_t qh_table_slots(const qh_geom& g) {
  return 1ull << (g.lsize + (g.all_shards ? g.shard_bits : 0));
}

// position of a key given its hash h = qh_mix(key)
QH_HD qh_pos qh_locate_hash(const qh_geom& g, uint64_t h) {
  const int gsize = g.lsize + g.shard_bits;
  qh_pos p;
  p.shard = (uint32_t)(h & qh_low_mask(g.shard_bits));
  p.home = (h >> g.shard_bits) & qh_low_mask(g.lsize);
  p.tag0 = (gsize >= 64 ? 0 : (h >> gsize)) << g.pbits;
  if (g.all_shards) {
    p.home |= (uint64_t)p.shard << g.lsize;
    p.shard = (uint32_t)g.shard_id;
  }
  return p;
}

// hash of a 2-bit packed (canonical) k-mer
QH_HD uint64_t qh_hash_key(uint64_t key, int k) { return qh_mix(qh_pkey(key, k), k); }

QH_HD qh_pos qh_locate(const qh_geom& g, uint64_t key) { return qh_locate_hash(g, qh_hash_key(key, g.k)); }

QH_HD uint64_t qh_probe_idx(const qh_geom& g, uint64_t home, uint32_t i) {
  const uint64_t wmask = qh_low_mask(g.wbits);
  return (home & ~wmask) | ((home + i) & wmask);
}

// hash of the key stored in the occupied slot word `slot` found at index `idx`
QH_HD uint64_t qh_slot_hash(const qh_geom& g, uint64_t slot, uint64_t idx) {
  const uint64_t wmask = qh_low_mask(g.wbits);
  const uint64_t f = slot >> g.vbits;
  const uint32_t i = (uint32_t)(f & qh_low_mask(g.pbits)) - 1;
  const uint64_t rem = f >> g.pbits;
  const uint64_t home = (idx & ~wmask) | ((idx - i) & wmask);
  const int gsize = g.lsize + g.shard_bits;
  const uint64_t shard = g.all_shards ? home >> g.lsize : (uint64_t)g.shard_id;
  uint64_t h = ((home & qh_low_mask(g.lsize)) << g.shard_bits) | shard;
  if (gsize < 64) h |= rem << gsize;
  return h & g.hmask;
}

// key stored in the occupied slot word `slot` found at index `idx`
QH_HD uint64_t qh_slot_key(const qh_geom& g, uint64_t slot, uint64_t idx) {
  return qh_key_from_pkey(qh_unmix(qh_slot_hash(g, slot, idx), g.k), g.k);
}

// value (count << 1 | hq) of the occupied slot word `slot` found at index `idx`
QH_HD uint64_t qh_slot_val(const qh_geom& g, uint64_t slot, uint64_t idx) {
  return g.wide ? g.wvals[idx] : slot & g.vmask;
}

// Quorum's update rule (src/mer_database.hpp:104-110) on a value field.
// Returns true when the field must change and writes the new field to *nv.
QH_HD bool qh_update_val(uint64_t v, uint32_t q, uint64_t max_val, uint64_t* nv) {
  if ((v & 1) < q) { *nv = 3; return true; }
  if ((v >> 1) == max_val || (v & 1) > q) return false;
  *nv = v + 2;
  return true;
}
