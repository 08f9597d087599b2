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
// returns the total to every thread. Thread t owns the span [t * per, (t+1) * per): spans of up to
// 32 values are held in registers so that their loads are all in flight together (a dependent
// loop of global loads made these one-workgroup kernels cost 35 us).
__device__ __forceinline__ uint32_t block_scan_u32(uint32_t* __restrict__ v, uint32_t n, uint32_t* part) {
  const uint32_t per = (n + 1023) / 1024;
  const uint32_t b = threadIdx.x * per;
  uint32_t s = 0;
  uint32_t c[32];
  const bool small = per <= 32;
  if (small) {
#pragma unroll
    for (int i = 0; i < 32; ++i) c[i] = ((uint32_t)i < per && b + i < n) ? v[b + i] : 0u;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += c[i];
  } else {
    for (uint32_t i = b; i < b + per && i < n; ++i) s += v[i];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (uint32_t o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan over the 1024 partials
    const uint32_t x = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += x;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - s;  // exclusive prefix of this thread's span
  if (small) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if ((uint32_t)i < per && b + i < n) v[b + i] = run;
      run += c[i];
    }
  } else {
    for (uint32_t i = b; i < b + per && i < n; ++i) {
      const uint32_t x = v[i];
      v[i] = run;
      run += x;
    }
  }
  return part[1023];
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

// ---------------------------------------------------------------- slot insertion

// hash_with_quality::add on the single-word slot table. Returns false when the
// reprobe limit is exhausted (caller queues the item for after the doubling).
__device__ __forceinline__ bool qh_insert(const qh_geom& g, uint64_t* __restrict__ table, uint64_t home,
                                          uint64_t tag0, uint32_t q, uint32_t* is_new) {
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

