// qh_device.h -- device-side building blocks shared by the kernels of
// qh_kernels.hip (direct global-CAS build, read side) and qh_partition.hip
// (radix-partitioned LDS-window build): wave helpers, the slot insertion of
// hash_with_quality::add (src/mer_database.hpp:94-113) and the wave-parallel
// rolling canonical k-mer extraction of src/create_database.cc:64-92.
//
// Internal item format everywhere below the C ABI: (h << 1) | q with
// h = qh_mix(qh_pkey(canonical mer)) -- the hash is a bijection of the 2k-bit key, so
// nothing is lost and no kernel after extraction recomputes it.
#pragma once
#include <hip/hip_runtime.h>
#include "qh_kernels.h"

// ------------------------------------------------------------------ small helpers

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {
  // number of set bits of `mask` in lanes below the calling lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// Exclusive scan of v[0, n) in place by one workgroup of 1024 threads (part = 1024 words of LDS);
// returns the total to every thread. The array passes through LDS 32 K values at a time, so that the global
// accesses are coalesced and all in flight together (a dependent loop of global loads made these one-workgroup
// kernels cost 35 us for 27 K values, span-strided loads 22 us, staged 10 us; the 165 K tile counts of a 2.7 GB
// chunk took 0.3-0.5 ms as spans of 162 values per thread, 0.06 ms as six staged rounds). Within a round thread t
// owns the span [t * per, (t+1) * per).
__device__ __forceinline__ uint32_t block_scan_u32(uint32_t* __restrict__ v, uint32_t n, uint32_t* part) {
  constexpr uint32_t ROUND = 32 * 1024;
  __shared__ uint32_t stage[ROUND];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < n; base += ROUND) {
    uint32_t* __restrict__ vr = v + base;
    const uint32_t m = n - base < ROUND ? n - base : ROUND;
    const uint32_t per = (m + 1023) / 1024;  // <= 32
    const uint32_t b = threadIdx.x * per;
    uint32_t s = 0;
    uint32_t c[32];
    {
      uint32_t in[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const uint32_t idx = (uint32_t)i * 1024 + threadIdx.x;
        in[i] = ((uint32_t)i < per && idx < m) ? vr[idx] : 0u;
      }
#pragma unroll
      for (int i = 0; i < 32; ++i)
        if ((uint32_t)i < per) stage[(uint32_t)i * 1024 + threadIdx.x] = in[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; ++i) c[i] = ((uint32_t)i < per && b + i < m) ? stage[b + i] : 0u;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += c[i];
    // scan of the 1024 partials: inside each wave by shuffles, then the 16 wave totals through LDS
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(inc, o, 64);
      if (lane >= o) inc += y;
    }
    if (lane == 63) part[wave] = inc;
    __syncthreads();  // also: every thread has read its span of stage[]
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const uint32_t y = part[w];
      if (w < wave) wbase += y;
      total += y;
    }
    uint32_t run = carry + wbase + inc - s;  // exclusive prefix of this thread's span
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if ((uint32_t)i < per && b + i < m) stage[b + i] = run;
      run += c[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const uint32_t idx = (uint32_t)i * 1024 + threadIdx.x;
      if ((uint32_t)i < per && idx < m) vr[idx] = stage[idx];
    }
    carry += total;
    __syncthreads();  // stage[] and part[] are free for the next round
  }
  return carry;
}

// ---- exclusive scan of a long u32 array by many small workgroups (three launches: block sums, their scan, apply).
// The one-workgroup scan above keeps 128 KiB of LDS, which no CU has to spare while a partition or window kernel of
// another chunk runs there: the preparation of the next chunk would wait for that kernel to end. These kernels keep a
// few bytes of LDS each. `sums` is scratch of ceil(n / QS_BLOCK) + 1 words.
#define QS_BLOCK 2048u   // values per workgroup of 256 threads
#define QS_MAX_BLOCKS 1024u

// n_eff = min(n, ceil(*bound / bound_div)) when bound != nullptr (values past it are zero and nobody reads their prefix)
__device__ __forceinline__ uint32_t qs_n_eff(uint32_t n, const uint32_t* bound, uint32_t bound_div) {
  if (!bound) return n;
  const uint32_t used = (*bound + bound_div - 1) / bound_div;
  return used < n ? used : n;
}

__device__ __forceinline__ void qs_block_sums(const uint32_t* __restrict__ v, uint32_t n, uint32_t* __restrict__ sums) {
  __shared__ uint32_t ws[4];
  const uint32_t base = blockIdx.x * QS_BLOCK;
  uint32_t t = 0;
  for (uint32_t i = threadIdx.x; i < QS_BLOCK; i += 256) t += base + i < n ? v[base + i] : 0u;
  t = wave_sum_u32(t);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// one workgroup of 1024 threads: sums[0, nb) -> their exclusive prefix; returns the total to every thread
__device__ __forceinline__ uint32_t qs_top(uint32_t* __restrict__ sums, uint32_t nb) {
  __shared__ uint32_t wt[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t x = threadIdx.x < nb ? sums[threadIdx.x] : 0u;
  uint32_t inc = x;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(inc, o, 64);
    if (lane >= o) inc += y;
  }
  if (lane == 63) wt[wave] = inc;
  __syncthreads();
  uint32_t wbase = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const uint32_t y = wt[w];
    if (w < wave) wbase += y;
    total += y;
  }
  if (threadIdx.x < nb) sums[threadIdx.x] = wbase + inc - x;
  return total;
}

// block b of v: exclusive scan in place, offset by sums[b] (thread t owns values [8t, 8t + 8) of the block)
__device__ __forceinline__ void qs_apply(uint32_t* __restrict__ v, uint32_t n, const uint32_t* __restrict__ sums) {
  __shared__ uint32_t wt[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t base = blockIdx.x * QS_BLOCK + threadIdx.x * 8u;
  uint32_t c[8], s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { c[i] = base + i < n ? v[base + i] : 0u; s += c[i]; }
  uint32_t inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(inc, o, 64);
    if (lane >= o) inc += y;
  }
  if (lane == 63) wt[wave] = inc;
  __syncthreads();
  uint32_t run = sums[blockIdx.x] + inc - s;
  for (int w = 0; w < wave; ++w) run += wt[w];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (base + i < n) v[base + i] = run;
    run += c[i];
  }
}

// 16-bit mask of the bytes of v that equal '\n'
__device__ __forceinline__ uint32_t nl_mask16(uint4 v) {
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t x = w[i] ^ 0x0A0A0A0Au;
    const uint32_t t = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);  // 0x80 at zero bytes
    const uint32_t c = ((t >> 7) & 1u) | ((t >> 14) & 2u) | ((t >> 21) & 4u) | ((t >> 28) & 8u);
    m |= c << (4 * i);
  }
  return m;
}

// 16 bytes at off (off is a multiple of 16; buf is 16-byte aligned); bytes past len read as 0
__device__ __forceinline__ uint4 load16_guarded(const uint8_t* __restrict__ buf, size_t off, size_t len) {
  if (off + 16 <= len) return *reinterpret_cast<const uint4*>(buf + off);
  uint32_t w[4] = {0, 0, 0, 0};
  for (int i = 0; i < 16; ++i)
    if (off + i < len) w[i >> 2] |= (uint32_t)buf[off + i] << (8 * (i & 3));
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// SWAR over 4 bytes of text at a time: zero-byte tests on u ^ 'A' give validity, two shifts the
// code, a biased subtract the signed quality compare.
__device__ __forceinline__ uint32_t swar_zero_bytes(uint32_t x) {  // 0x80 where a byte of x is 0
  return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
}

struct qh_swar_thr { uint32_t low7; bool neg; };  // the quality threshold ^ 0x80: low 7 bits replicated, top bit
__device__ __forceinline__ qh_swar_thr qh_swar_threshold(int min_qual) {
  const uint32_t tb = ((uint32_t)(int8_t)min_qual ^ 0x80u) & 0xFFu;
  return qh_swar_thr{(tb & 0x7Fu) * 0x01010101u, (tb & 0x80u) != 0};
}

// flags of 4 bytes: bit 0 valid, 1 hq, 2 code bit 0, 3 code bit 1 of each byte
__device__ __forceinline__ uint32_t swar_byte_flags(uint32_t c, const qh_swar_thr& thr) {
  const uint32_t u = c & 0xDFDFDFDFu, e = u ^ 0x41414141u;
  const uint32_t z = swar_zero_bytes(e & 0xFDFDFDFDu) | swar_zero_bytes(e ^ 0x06060606u) | swar_zero_bytes(e ^ 0x15151515u);
  const uint32_t vm = (z - (z >> 7)) | z;  // 0xFF per valid byte
  const uint32_t code2 = ((u << 1) ^ u) & 0x0C0C0C0Cu;
  // signed c >= t  <=>  unsigned (c ^ 0x80) >= (t ^ 0x80); thr_low7 / thr_neg describe t ^ 0x80
  const uint32_t a = c ^ 0x80808080u;
  const uint32_t r = (a | 0x80808080u) - thr.low7;  // bit 7: low 7 bits of a >= low 7 bits of the threshold
  const uint32_t ge = thr.neg ? (a & r) : (a | r);
  return ((code2 | 0x01010101u) & vm) | ((ge >> 6) & 0x02020202u);
}

// flags of 8 bytes (two dwords) -> byte p = plane p (v, h, l0, l1) of the 8 bytes, by two delta swaps
__device__ __forceinline__ uint32_t swar_transpose8(uint32_t f_lo, uint32_t f_hi) {
  uint32_t g = f_lo | (f_hi << 4);
  uint32_t t = ((g >> 7) ^ g) & 0x00AA00AAu;
  g ^= t ^ (t << 7);
  t = ((g >> 14) ^ g) & 0x0000CCCCu;
  g ^= t ^ (t << 14);
  return g;
}

// 16 bytes of chunk text (unit u) -> 16 bits of each text plane (v, h, l0, l1; see qh_partition.hip)
__device__ __forceinline__ void qh_store_text_planes(const uint4& x, size_t u, uint16_t* __restrict__ planes,
                                                     size_t plane_halfwords, const qh_swar_thr& thr) {
  const uint32_t g0 = swar_transpose8(swar_byte_flags(x.x, thr), swar_byte_flags(x.y, thr));
  const uint32_t g1 = swar_transpose8(swar_byte_flags(x.z, thr), swar_byte_flags(x.w, thr));
  // halfword of plane p = byte p of g0 | byte p of g1 << 8
  const uint32_t vh = __builtin_amdgcn_perm(g1, g0, 0x05010400u);  // {g0.b0, g1.b0, g0.b1, g1.b1}
  const uint32_t ll = __builtin_amdgcn_perm(g1, g0, 0x07030602u);  // {g0.b2, g1.b2, g0.b3, g1.b3}
  planes[u] = (uint16_t)vh;
  planes[plane_halfwords + u] = (uint16_t)(vh >> 16);
  planes[2 * plane_halfwords + u] = (uint16_t)ll;
  planes[3 * plane_halfwords + u] = (uint16_t)(ll >> 16);
}

// The same flags by table: what swar_byte_flags computes for a byte, plus bit 4 = the byte is a newline. 256 entries
// in LDS, filled once per workgroup (qh_byte_flags_of); a dword of text is four ds_read_u8 instead of ~50 VALU
// instructions, which is what bounds the one kernel that reads all of the text.
__device__ __forceinline__ uint32_t qh_byte_flags_of(uint32_t c, int min_qual) {
  const uint32_t u = c & 0xDFu;
  const uint32_t code = u == 'A' ? 0u : u == 'C' ? 1u : u == 'G' ? 2u : u == 'T' ? 3u : 4u;
  uint32_t f = code < 4u ? (1u | (code << 2)) : 0u;
  if ((int)(int8_t)c >= (int)(int8_t)min_qual) f |= 2u;
  if (c == '\n') f |= 16u;
  return f;
}
__device__ __forceinline__ uint32_t qh_lut_flags4(const uint8_t* lut, uint32_t w) {
  return (uint32_t)lut[w & 0xFFu] | ((uint32_t)lut[(w >> 8) & 0xFFu] << 8) | ((uint32_t)lut[(w >> 16) & 0xFFu] << 16) |
         ((uint32_t)lut[w >> 24] << 24);
}
// flags of the unit's 16 bytes (four dwords from qh_lut_flags4) -> its 16 newline bits
__device__ __forceinline__ uint32_t qh_nl_from_flags(const uint32_t (&f)[4]) {
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t t = f[i] >> 4;
    m |= ((t & 1u) | ((t >> 7) & 2u) | ((t >> 14) & 4u) | ((t >> 21) & 8u)) << (4 * i);
  }
  return m;
}
__device__ __forceinline__ void qh_store_text_planes_flags(const uint32_t (&f)[4], size_t u, uint16_t* __restrict__ planes,
                                                           size_t plane_halfwords) {
  const uint32_t g0 = swar_transpose8(f[0] & 0x0F0F0F0Fu, f[1] & 0x0F0F0F0Fu);
  const uint32_t g1 = swar_transpose8(f[2] & 0x0F0F0F0Fu, f[3] & 0x0F0F0F0Fu);
  const uint32_t vh = __builtin_amdgcn_perm(g1, g0, 0x05010400u);
  const uint32_t ll = __builtin_amdgcn_perm(g1, g0, 0x07030602u);
  planes[u] = (uint16_t)vh;
  planes[plane_halfwords + u] = (uint16_t)(vh >> 16);
  planes[2 * plane_halfwords + u] = (uint16_t)ll;
  planes[3 * plane_halfwords + u] = (uint16_t)(ll >> 16);
}

// the same into a tile's staging area: plane p of unit u (0..1023 inside the tile) at stage[p * 1024 + u]
__device__ __forceinline__ void qh_stage_text_planes_flags(const uint32_t (&f)[4], uint32_t u, uint16_t* stage) {
  const uint32_t g0 = swar_transpose8(f[0] & 0x0F0F0F0Fu, f[1] & 0x0F0F0F0Fu);
  const uint32_t g1 = swar_transpose8(f[2] & 0x0F0F0F0Fu, f[3] & 0x0F0F0F0Fu);
  const uint32_t vh = __builtin_amdgcn_perm(g1, g0, 0x05010400u);
  const uint32_t ll = __builtin_amdgcn_perm(g1, g0, 0x07030602u);
  stage[u] = (uint16_t)vh;
  stage[1024 + u] = (uint16_t)(vh >> 16);
  stage[2048 + u] = (uint16_t)ll;
  stage[3072 + u] = (uint16_t)(ll >> 16);
}

// ---------------------------------------------------------------- slot insertion

// hash_with_quality::add on the single-word slot table. Returns false when the
// reprobe limit is exhausted (caller queues the item for after the doubling).
// Wide form (qh_spec.h): the slot holds the key only and is claimed first; the value is then updated in the side
// array by the compare-and-swap loop of mer_database.hpp:102-111 itself -- two dependent random accesses, as upstream.
__device__ __noinline__ bool qh_insert_wide(const qh_geom& g, uint64_t* __restrict__ table, uint64_t home,
                                            uint64_t tag0, uint32_t q, uint32_t* is_new) {
  const uint64_t wmask = qh_low_mask(g.wbits);
  const uint64_t base = home & ~wmask;
  for (uint32_t i = 0; i <= (uint32_t)g.max_reprobe; ++i) {
    const uint64_t idx = base | ((home + i) & wmask);
    unsigned long long* slot = reinterpret_cast<unsigned long long*>(table + idx);
    const uint64_t want = tag0 | (i + 1);
    uint64_t old = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == 0) {
      old = atomicCAS(slot, 0ull, (unsigned long long)want);
      if (old == 0) { *is_new = 1; old = want; }
    }
    if (old != want) continue;  // another key lives here: next probe
    unsigned long long* vp = reinterpret_cast<unsigned long long*>(g.wvals + idx);
    uint64_t v = __hip_atomic_load(vp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
      uint64_t nv;
      if (!qh_update_val(v, q, g.max_val, &nv)) return true;
      const uint64_t prev = atomicCAS(vp, (unsigned long long)v, (unsigned long long)nv);
      if (prev == v) return true;
      v = prev;
    }
  }
  return false;
}

__device__ __forceinline__ bool qh_insert(const qh_geom& g, uint64_t* __restrict__ table, uint64_t home,
                                          uint64_t tag0, uint32_t q, uint32_t* is_new) {
  if (g.wide) return qh_insert_wide(g, table, home, tag0, q, is_new);
  const uint64_t wmask = qh_low_mask(g.wbits);
  const uint64_t base = home & ~wmask;
  const uint64_t fresh = q ? 3u : 2u;  // nval of a new entry (mer_database.hpp:105-110 from 0)
  for (uint32_t i = 0; i <= (uint32_t)g.max_reprobe; ++i) {
    unsigned long long* slot = reinterpret_cast<unsigned long long*>(table + (base | ((home + i) & wmask)));
    const uint64_t want = tag0 | (i + 1);
    uint64_t old = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
      if (old == 0) {
        const uint64_t prev = atomicCAS(slot, 0ull, (unsigned long long)((want << g.vbits) | fresh));
        if (prev == 0) { *is_new = 1; return true; }
        old = prev;
      }
      if ((old >> g.vbits) != want) break;  // another key lives here: next probe
      uint64_t nv;
      if (!qh_update_val(old & g.vmask, q, g.max_val, &nv)) return true;
      const uint64_t nw = (old & ~g.vmask) | nv;
      const uint64_t prev = atomicCAS(slot, (unsigned long long)old, (unsigned long long)nw);
      if (prev == old) return true;
      old = prev;
    }
  }
  return false;
}

__device__ __forceinline__ void qh_overflow(qh_meta* __restrict__ meta, uint64_t* __restrict__ ovf,
                                            uint64_t ovf_cap, uint64_t item) {
  const unsigned long long at = atomicAdd(reinterpret_cast<unsigned long long*>(&meta->ovf_count), 1ull);
  if (at < ovf_cap) ovf[at] = item;
  else meta->ovf_lost = 1;
}

// Wave-parallel form of create_database.cc:64-92 for one 64-base column block of
// one read. Each lane owns one base; the wave's four ballots are the bit planes
// (valid, high-quality, code bit 0, code bit 1) of 64 consecutive positions, and
// the k-mer ending at a lane's base is the k-bit window ending at its bit of the
// (previous block : this block) plane pair. "low_len >= k" is "valid window all
// ones", "high_len >= k" is "hq window all ones": the registers m/rm of the
// scalar loop never need clearing because only complete windows are emitted.
struct qh_planes { uint64_t v, h, l0, l1; };

// the 32 positions ending at this lane's base: bit 31 = the lane's position, bit 31-i = i earlier
__device__ __forceinline__ uint32_t window_top32(uint64_t cur, uint64_t prev, int lane) {
  return (uint32_t)(((cur << (63 - lane)) | ((prev >> lane) >> 1)) >> 32);
}

// min(m, rm) of create_database.cc:86 on plane forms (m0/m1 = low/high code bits of the forward
// mer, r0/r1 of its reverse complement, base i at bit i), returned as a PLANE KEY (qh_pkey form).
// m < rm as 2-bit packed integers is decided at the most significant base where they differ:
// with that base shifted to bit 31, forward wins iff its high bit is 0 where the high bits
// differ, else iff its low bit is 0.
__device__ __forceinline__ void qh_canonical_planes(uint32_t m0, uint32_t m1, uint32_t r0, uint32_t r1, uint32_t* c0,
                                                    uint32_t* c1) {
  const uint32_t d1 = m1 ^ r1, d = (m0 ^ r0) | d1;
  const int c = __clz((int)d) & 31;  // d == 0 (palindrome): either
  const uint32_t sel = ((int32_t)(d1 << c) < 0) ? m1 : m0;
  const bool fwd = (int32_t)(sel << c) >= 0;
  *c0 = fwd ? m0 : r0;
  *c1 = fwd ? m1 : r1;
}

// Hash (qh_mix, qh_spec.h) of the canonical form min(m, r) of a k-mer whose two strands are at hand as
// planes: m = forward, r = reverse complement. Same value as qh_mix_pair(canonical, other), with the
// two comparisons -- whole mers for the key (create_database.cc:86), cores for the home slot -- sharing
// their XORs and made in place (the core is bits 1..k-2 of the planes).
// SMALLK: k < 4 (no core: the whole key is hashed); callers in a per-item loop pick the form once
template <bool SMALLK>
__device__ __forceinline__ uint64_t qh_hash_strands_form(uint32_t m0, uint32_t m1, uint32_t r0, uint32_t r1, int k) {
  // m < r is decided at the most significant base where they differ. lt marks the bases where the forward code is
  // the smaller one, d the bases that differ at all: lt and d ^ lt are disjoint, so the one that holds the top bit
  // of d is the larger integer -- no count-leading-zeros, no variable shifts. d == 0 (palindrome): either strand.
  const uint32_t x1 = m1 ^ r1, d = (m0 ^ r0) | x1;
  const uint32_t lt = (r1 & ~m1) | (~x1 & r0 & ~m0);
  const bool fwd = lt > (d ^ lt);
  const uint32_t c0 = fwd ? m0 : r0, c1 = fwd ? m1 : r1;
  if (SMALLK) return qh_feistel3(c0, c1, k);
  const int kc = k - 2;
  const uint32_t cm = (1u << kc) - 1u;
  const uint32_t dc = d & (cm << 1), ltc = lt & (cm << 1);
  // the strand with the smaller core supplies it; equal cores: the key's own strand (no flip)
  const bool cfwd = (ltc > (dc ^ ltc)) | ((dc == 0) & fwd);
  const uint32_t cL = ((cfwd ? m0 : r0) >> 1) & cm, cR = ((cfwd ? m1 : r1) >> 1) & cm;
  // e = c0 bit 0 | c1 bit 0 << 1 | c0 bit k-1 << 2 | c1 bit k-1 << 3 (qh_mix_pair), the two planes' end bits gathered together
  const uint32_t em = 1u | (1u << (k - 1));
  const uint32_t u = (c0 & em) | ((c1 & em) << 1);
  const uint32_t e = (u & 3u) | ((u >> (k - 3)) & 12u);
  return (qh_feistel3(cL, cR, kc) ^ QH_MIX_Z) | ((uint64_t)(e | ((uint32_t)(fwd != cfwd) << 4)) << (2 * kc));
}

__device__ __forceinline__ uint64_t qh_hash_strands(uint32_t m0, uint32_t m1, uint32_t r0, uint32_t r1, int k) {
  return k < 4 ? qh_hash_strands_form<true>(m0, m1, r0, r1, k) : qh_hash_strands_form<false>(m0, m1, r0, r1, k);
}

__device__ __forceinline__ uint64_t qh_canonical_pkey(uint32_t m0, uint32_t m1, uint32_t r0, uint32_t r1, int k) {
  uint32_t c0, c1;
  qh_canonical_planes(m0, m1, r0, r1, &c0, &c1);
  return (uint64_t)c0 | ((uint64_t)c1 << k);
}

// Emits the canonical k-mer ending at this lane's base as a PLANE KEY (qh_pkey form: low code
// bits | high code bits << k); the 2-bit interleaved form is never built on this path.
__device__ __forceinline__ bool qh_lane_kmer(const qh_planes& cur, const qh_planes& prev, int lane, int k,
                                             uint64_t* pkey, uint32_t* hq) {
  const uint32_t kones = (uint32_t)((1ull << k) - 1);
  const int drop = 32 - k;
  if ((window_top32(cur.v, prev.v, lane) >> drop) != kones) return false;
  *hq = (window_top32(cur.h, prev.h, lane) >> drop) == kones;
  const uint32_t x0 = window_top32(cur.l0, prev.l0, lane), x1 = window_top32(cur.l1, prev.l1, lane);
  // forward mer: base i (i = 0 newest) is window bit 31-i -> bit-reverse
  const uint32_t m0 = __builtin_bitreverse32(x0) & kones, m1 = __builtin_bitreverse32(x1) & kones;
  // reverse complement: base i = 3 - (forward base k-1-i) -> complemented window, oldest base lowest
  const uint32_t r0 = (~x0) >> drop, r1 = (~x1) >> drop;
  *pkey = qh_canonical_pkey(m0, m1, r0, r1, k);
  return true;
}

__device__ __forceinline__ qh_planes qh_load_planes(const uint8_t* __restrict__ buf, uint32_t s0, uint32_t q0,
                                                    uint32_t len, uint32_t c0, int lane, int min_qual) {
  const uint32_t j = c0 + lane;
  const bool inb = j < len;
  const unsigned b = inb ? buf[(size_t)s0 + j] : 0u;
  const int ql = inb ? (int)(int8_t)buf[(size_t)q0 + j] : -128;
  const int code = qh_code(b);
  const bool valid = code >= 0;
  qh_planes p;
  p.v = __ballot(valid);
  p.h = __ballot(valid && ql >= (int)(int8_t)min_qual);  // signed char compare, create_database.cc:81
  p.l0 = __ballot(valid && (code & 1));
  p.l1 = __ballot(valid && (code & 2));
  return p;
}

