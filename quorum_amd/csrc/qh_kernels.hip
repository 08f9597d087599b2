// qh_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of libquorum_hip.
//
// Hot path replaced (paths relative to the Quorum tree):
//   FASTQ chunk -> records            jellyfish::whole_sequence_parser, src/create_database.cc:52,61-66
//   per-base rolling canonical k-mer  src/create_database.cc:64-92
//   hash_with_quality::add            src/mer_database.hpp:94-113
//   table doubling / rehash           src/mer_database.hpp:137-187
//   database_query lookups            src/mer_database.hpp:284-329
//   table reductions                  src/error_correct_reads.cc:650-659, src/histo_mer_database.cc:13-21
//
// Everything here is integer/byte work bound by HBM (random 8-byte slot updates
// plus a streaming pass over the chunk); there is no MFMA-shaped work.
#include "qh_kernels.h"
#include "qh_device.h"

#define QH_BLOCK 256

// ------------------------------------------------------- FASTQ chunk -> record table

// pass 1, the only pass over the chunk's text: newlines per 16 KiB tile, the tile's newlines as a compact list for
// pass 3 -- one 16-bit entry each: {byte position inside the tile : 14 | the next byte is '@' : 1 | is '+' : 1}, in
// text order, QH_NL_CAP to a tile -- and, when the chunk goes through the partitioned build, the four text planes of
// qh_partition.hip. (Round 1 wrote a dense plane of newline bits, 8 bytes per 16 bytes of text whether they held a
// newline or not: 0.22 GB written here and read back by pass 3 for the 16 MB of entries they carried.)
#define QH_NL_CAP 4096u  // more newlines in 16 KiB: lines of under 4 bytes, records the fast path rejects anyway
__global__ __launch_bounds__(QH_BLOCK) void k_index_text(const uint8_t* __restrict__ buf, size_t len,
                                                         uint32_t* __restrict__ tile_counts,
                                                         uint16_t* __restrict__ nl_list,
                                                         uint16_t* __restrict__ planes, size_t plane_halfwords,
                                                         int min_qual, qh_meta* __restrict__ meta) {
  constexpr uint32_t UNITS = QH_TILE_BYTES / 16;  // 1024: unit j * 256 + t is thread t's j-th
  __shared__ uint32_t ucnt[UNITS];   // newlines per unit, then their exclusive prefix in text order
  __shared__ __attribute__((aligned(16))) uint16_t stage[QH_NL_CAP];  // the tile's planes (4 x 1024 halfwords), then its newline entries
  static_assert(QH_NL_CAP == 4096, "stage holds four planes of a tile");
  __shared__ uint32_t wsum[QH_BLOCK / 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const size_t tile_base = (size_t)blockIdx.x * QH_TILE_BYTES;
  __shared__ uint8_t lut[256];  // flags of every byte value (QH_BLOCK == 256 threads fill it)
  static_assert(QH_BLOCK == 256, "one table entry per thread");
  lut[tid] = (uint8_t)qh_byte_flags_of((uint32_t)tid, min_qual);
  __syncthreads();
  uint32_t m[4], m_at[4], m_plus[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t u = (uint32_t)j * QH_BLOCK + tid;
    const size_t off = tile_base + (size_t)u * 16;
    m[j] = 0; m_at[j] = 0; m_plus[j] = 0;
    if (off < len) {
      const uint4 x = load16_guarded(buf, off, len);
      const uint32_t fl[4] = {qh_lut_flags4(lut, x.x), qh_lut_flags4(lut, x.y), qh_lut_flags4(lut, x.z), qh_lut_flags4(lut, x.w)};
      m[j] = qh_nl_from_flags(fl);
      // what follows each newline: '@' (a header must) or '+' (a separator must); pass 3 knows which
      const uint32_t w[5] = {x.x, x.y, x.z, x.w, off + 16 < len ? (uint32_t)buf[off + 16] : 0u};
      uint32_t mm = m[j];
      while (mm) {
        const int b = __ffs(mm) - 1;
        mm &= mm - 1;
        const int nb = b + 1, ws = nb >> 2;
        const uint32_t d = ws == 0 ? w[0] : ws == 1 ? w[1] : ws == 2 ? w[2] : ws == 3 ? w[3] : w[4];
        const uint32_t c = (d >> (8 * (nb & 3))) & 0xFFu;
        m_at[j] |= (uint32_t)(c == '@') << b;
        m_plus[j] |= (uint32_t)(c == '+') << b;
      }
      if (planes) qh_stage_text_planes_flags(fl, u, stage);
    } else if (planes) {
      stage[u] = 0; stage[1024 + u] = 0; stage[2048 + u] = 0; stage[3072 + u] = 0;
    }
    ucnt[u] = __popc(m[j]);
  }
  __syncthreads();
  if (planes) {
    // the tile's planes leave in 16-byte pieces (2 per thread) instead of sixteen 2-byte stores per thread: plane p of the
    // tile is 2 KiB in a row, 16-byte aligned (plane_halfwords is a multiple of 8, a tile 1024 halfwords)
    const uint4* st4 = reinterpret_cast<const uint4*>(stage);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const uint32_t c = (uint32_t)r * QH_BLOCK + tid, p = c >> 7, w = c & 127u;
      *reinterpret_cast<uint4*>(planes + (size_t)p * plane_halfwords + (size_t)blockIdx.x * 1024 + (size_t)w * 8) = st4[c];
    }
    __syncthreads();  // stage is reused for the newline entries
  }
  // exclusive prefix of the units' counts in text order: thread t scans units 4t .. 4t+3 (measured: prefix sums by
  // shuffles over a (wave, j, lane) layout instead of this pass through LDS cost the step 0.02 ms)
  uint32_t c4[4], cs = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { c4[i] = ucnt[4 * tid + i]; cs += c4[i]; }
  uint32_t inc = cs;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  uint32_t wbase = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < QH_BLOCK / 64; ++w) {
    if (w < wv) wbase += wsum[w];
    tot += wsum[w];
  }
  {
    uint32_t run = wbase + inc - cs;
#pragma unroll
    for (int i = 0; i < 4; ++i) { ucnt[4 * tid + i] = run; run += c4[i]; }
  }
  if (tid == 0) {
    tile_counts[blockIdx.x] = tot;
    if (tot > QH_NL_CAP) { atomicOr(&meta->format_err, QH_FMT_TOO_MANY); atomicMin(&meta->first_bad, 0u); }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t u = (uint32_t)j * QH_BLOCK + tid;
    uint32_t idx = ucnt[u], mm = m[j];
    while (mm) {
      const int b = __ffs(mm) - 1;
      mm &= mm - 1;
      if (idx < QH_NL_CAP)
        stage[idx] = (uint16_t)((u * 16u + (uint32_t)b) | (((m_at[j] >> b) & 1u) << 14) | (((m_plus[j] >> b) & 1u) << 15));
      ++idx;
    }
  }
  __syncthreads();
  const uint32_t nst = tot < QH_NL_CAP ? tot : QH_NL_CAP;
  uint16_t* __restrict__ dst = nl_list + (size_t)blockIdx.x * QH_NL_CAP;
  for (uint32_t i = tid; i < nst; i += QH_BLOCK) dst[i] = stage[i];
}

// pass 2: exclusive scan of the tile counts (block sums, their scan with the chunk-level checks, apply: small
// workgroups that fit beside another chunk's partition kernel, see qh_device.h)
__global__ __launch_bounds__(256) void k_scan_tiles_sums(const uint32_t* __restrict__ tile_counts, uint32_t ntiles,
                                                         uint32_t* __restrict__ sums) {
  qs_block_sums(tile_counts, ntiles, sums);
}

__global__ __launch_bounds__(1024) void k_scan_tiles_top(uint32_t* __restrict__ tile_counts, uint32_t ntiles,
                                                         uint32_t* __restrict__ sums, uint32_t nb,
                                                         const uint8_t* __restrict__ buf, size_t len,
                                                         uint32_t rec_cap, qh_meta* __restrict__ meta) {
  const uint32_t total_lines = qs_top(sums, nb);
  if (threadIdx.x == 1023) {
    uint32_t lines = total_lines;
    if (len > 0 && buf[len - 1] != '\n') ++lines;  // last line without '\n'
    uint32_t err = 0;
    if (lines & 3u) err |= QH_FMT_LINES_MOD4;
    if ((lines >> 2) > rec_cap) err |= QH_FMT_TOO_MANY;
    if (len > 0 && buf[0] != '@') err |= QH_FMT_NO_AT;
    err |= meta->format_err;  // pass 1: a tile with more newlines than its list holds
    tile_counts[ntiles] = total_lines;  // pass 3 takes a tile's line count from the difference of two prefixes
    meta->n_lines = lines;
    meta->nrec = err ? 0 : (lines >> 2);
    meta->format_err = err;
    meta->first_bad = err ? 0 : 0xFFFFFFFFu;
  }
}

__global__ __launch_bounds__(256) void k_scan_tiles_apply(uint32_t* __restrict__ tile_counts, uint32_t ntiles,
                                                          const uint32_t* __restrict__ sums) {
  qs_apply(tile_counts, ntiles, sums);
}

// a newline at byte p ends line L: the matching field of record L/4 (word L of the record table, returned) and
// the check of the first byte of line L+1 ('@' for a header line, '+' for a separator); at / plus say what that
// byte is
__device__ __forceinline__ uint32_t line_end(size_t len, uint32_t L, uint32_t p, bool at, bool plus,
                                             qh_meta* __restrict__ meta) {
  if ((size_t)p + 1 < len) {
    const uint32_t nt = (L + 1) & 3u;
    if (nt == 0 && !at) { atomicOr(&meta->format_err, QH_FMT_NO_AT); atomicMin(&meta->first_bad, (L + 1) >> 2); }
    if (nt == 2 && !plus) { atomicOr(&meta->format_err, QH_FMT_NO_PLUS); atomicMin(&meta->first_bad, (L + 1) >> 2); }
  }
  return (L & 1u) ? p : p + 1;  // x=seq_off y=seq_end z=qual_off w=qual_end
}

// pass 3: every newline writes its field of the record table. The table, read as words, is indexed by the line
// number, a tile's lines are consecutive and its list is in text order: entry i of tile t is line prefix[t] + i, the
// loads and the stores are both contiguous runs.
__global__ __launch_bounds__(QH_BLOCK) void k_fill_records(const uint8_t* __restrict__ buf, size_t len,
                                                           const uint16_t* __restrict__ nl_list,
                                                           const uint32_t* __restrict__ tile_prefix,
                                                           uint32_t* __restrict__ recs, uint32_t rec_cap,
                                                           qh_meta* __restrict__ meta) {
  if (meta->format_err & (QH_FMT_LINES_MOD4 | QH_FMT_TOO_MANY)) return;
  const uint32_t L0 = tile_prefix[blockIdx.x];  // (the scan leaves the total behind the last tile)
  uint32_t n = tile_prefix[blockIdx.x + 1] - L0;
  if (n > QH_NL_CAP) n = QH_NL_CAP;  // never past the tile's list (pass 1 has rejected such a chunk)
  const uint64_t words = (uint64_t)rec_cap * 4;
  const uint16_t* __restrict__ src = nl_list + (size_t)blockIdx.x * QH_NL_CAP;
  const size_t tile_base = (size_t)blockIdx.x * QH_TILE_BYTES;
  for (uint32_t i = threadIdx.x; i < n; i += QH_BLOCK) {
    const uint32_t e = src[i];
    const uint32_t f = line_end(len, L0 + i, (uint32_t)(tile_base + (e & 0x3FFFu)), (e >> 14) & 1u, (e >> 15) & 1u, meta);
    if ((uint64_t)L0 + i < words) recs[(size_t)L0 + i] = f;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && len > 0 && buf[len - 1] != '\n') {
    const uint32_t L = meta->n_lines - 1;
    const uint32_t f = line_end(len, L, (uint32_t)len, true, true, meta);
    if (L < words) recs[L] = f;
  }
}

// pass 4: length checks; rewrites each record to {seq_off, len, qual_off, 0}
__global__ __launch_bounds__(QH_BLOCK) void k_validate_records(uint4* __restrict__ recs, qh_meta* __restrict__ meta) {
  const uint32_t nrec = meta->nrec;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nrec; r += gridDim.x * blockDim.x) {
    const uint4 rc = recs[r];
    const uint32_t slen = rc.y - rc.x, qlen = rc.w - rc.z;
    if (rc.y < rc.x || rc.w < rc.z || slen != qlen) {
      atomicOr(&meta->format_err, QH_FMT_LEN_MISMATCH);
      atomicMin(&meta->first_bad, r);
    } else if (slen == 0) {  // upstream's parser does not consume the empty quality line
      atomicOr(&meta->format_err, QH_FMT_EMPTY_TAIL);
      atomicMin(&meta->first_bad, r);
    }
    recs[r] = make_uint4(rc.x, slen, rc.z, 0);
  }
}

void qh_launch_index_fastq(const uint8_t* buf, size_t len, uint32_t* tile_counts, ushort4* nl_plane, uint4* recs,
                           uint32_t rec_cap, qh_meta* meta, const qh_fastq_aux* planes, int min_qual, hipStream_t s) {
  const uint32_t ntiles = (uint32_t)((len + QH_TILE_BYTES - 1) / QH_TILE_BYTES);
  if (ntiles)
    hipLaunchKernelGGL(k_index_text, dim3(ntiles), dim3(QH_BLOCK), 0, s, buf, len, tile_counts,
                       reinterpret_cast<uint16_t*>(nl_plane),
                       planes ? reinterpret_cast<uint16_t*>(planes->planes) : nullptr,
                       planes ? (size_t)planes->plane_words * 4 : 0, min_qual, meta);
  // (tile_counts has room for ntiles + 1 counts and QH_SCAN_SCRATCH words of scratch behind them)
  uint32_t* sums = tile_counts + ntiles + 1;
  const uint32_t nb = (ntiles + QS_BLOCK - 1) / QS_BLOCK;
  if (nb) hipLaunchKernelGGL(k_scan_tiles_sums, dim3(nb), dim3(256), 0, s, tile_counts, ntiles, sums);
  hipLaunchKernelGGL(k_scan_tiles_top, dim3(1), dim3(1024), 0, s, tile_counts, ntiles, sums, nb, buf, len, rec_cap, meta);
  if (nb) hipLaunchKernelGGL(k_scan_tiles_apply, dim3(nb), dim3(256), 0, s, tile_counts, ntiles, sums);
  if (ntiles)
    hipLaunchKernelGGL(k_fill_records, dim3(ntiles), dim3(QH_BLOCK), 0, s, buf, len,
                       reinterpret_cast<const uint16_t*>(nl_plane), tile_counts, reinterpret_cast<uint32_t*>(recs), rec_cap, meta);
  hipLaunchKernelGGL(k_validate_records, dim3(1024), dim3(QH_BLOCK), 0, s, recs, meta);
}

// One wave per read (grid-stride). Direct build: every emitted k-mer is inserted
// with global 64-bit CAS.
__global__ __launch_bounds__(QH_BLOCK) void k_build_direct(const uint8_t* __restrict__ buf,
                                                           const uint4* __restrict__ recs,
                                                           qh_meta* __restrict__ meta, const qh_geom g,
                                                           uint64_t* __restrict__ table,
                                                           uint64_t* __restrict__ ovf, uint64_t ovf_cap) {
  if (meta->format_err) return;  // chunk rejected as a whole: nothing may be inserted
  const uint32_t nrec = meta->nrec;
  const int lane = threadIdx.x & 63;
  const uint32_t wave0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  uint32_t n_inst = 0, n_hq = 0, n_new = 0, n_foreign = 0;
  for (uint32_t r = __builtin_amdgcn_readfirstlane(wave0); r < nrec; r += nwaves) {
    const uint4 rc = recs[r];
    const uint32_t s0 = rc.x, len = rc.y, q0 = rc.z;
    qh_planes prev = {0, 0, 0, 0};
    for (uint32_t c0 = 0; c0 < len; c0 += 64) {
      const qh_planes cur = qh_load_planes(buf, s0, q0, len, c0, lane, g.min_qual);
      uint64_t key;
      uint32_t q;
      if (qh_lane_kmer(cur, prev, lane, g.k, &key, &q)) {
        const uint64_t h = qh_mix(key, g.k);
        const qh_pos p = qh_locate_hash(g, h);
        if (p.shard != (uint32_t)g.shard_id) {
          ++n_foreign;
        } else {
          ++n_inst;
          n_hq += q;
          uint32_t is_new = 0;
          if (!qh_insert(g, table, p.home, p.tag0, q, &is_new)) qh_overflow(meta, ovf, ovf_cap, (h << 1) | q);
          n_new += is_new;
        }
      }
      prev = cur;
    }
  }
  n_inst = wave_sum_u32(n_inst); n_hq = wave_sum_u32(n_hq);
  n_new = wave_sum_u32(n_new); n_foreign = wave_sum_u32(n_foreign);
  if (lane == 0) {
    if (n_inst) atomicAdd(reinterpret_cast<unsigned long long*>(&meta->instances), (unsigned long long)n_inst);
    if (n_hq) atomicAdd(reinterpret_cast<unsigned long long*>(&meta->hq_instances), (unsigned long long)n_hq);
    if (n_new) atomicAdd(reinterpret_cast<unsigned long long*>(&meta->new_keys), (unsigned long long)n_new);
    if (n_foreign) atomicAdd(reinterpret_cast<unsigned long long*>(&meta->foreign), (unsigned long long)n_foreign);
  }
}

void qh_launch_build_direct(const uint8_t* buf, const uint4* recs, qh_meta* meta, const qh_geom& g,
                            uint64_t* table, uint64_t* ovf, uint64_t ovf_cap, hipStream_t s) {
  hipLaunchKernelGGL(k_build_direct, dim3(2048), dim3(QH_BLOCK), 0, s, buf, recs, meta, g, table, ovf, ovf_cap);
}

// packed items, one lane per item: (mer << 1 | q) as the C ABI passes them, or the
// internal hashed form (qh_mix(mer) << 1 | q) of the overflow / spill lists.
// n_dev (optional): device-side item count (the spill list's), capped by n.
__global__ __launch_bounds__(QH_BLOCK) void k_insert_items(const uint64_t* __restrict__ items, uint64_t n,
                                                           const unsigned long long* __restrict__ n_dev,
                                                           qh_meta* __restrict__ meta, const qh_geom g,
                                                           uint64_t* __restrict__ table,
                                                           uint64_t* __restrict__ ovf, uint64_t ovf_cap,
                                                           int count_instances, int hashed) {
  if (n_dev) { const unsigned long long nd = *n_dev; n = nd < n ? nd : n; }
  const int lane = threadIdx.x & 63;
  uint32_t n_inst = 0, n_hq = 0, n_new = 0, n_foreign = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t it = items[i];
    const uint32_t q = (uint32_t)(it & 1);
    const uint64_t h = hashed ? ((it >> 1) & g.hmask) : qh_hash_key((it >> 1) & g.kmask, g.k);
    const qh_pos p = qh_locate_hash(g, h);
    if (p.shard != (uint32_t)g.shard_id) { ++n_foreign; continue; }
    ++n_inst; n_hq += q;
    uint32_t is_new = 0;
    if (!qh_insert(g, table, p.home, p.tag0, q, &is_new)) qh_overflow(meta, ovf, ovf_cap, (h << 1) | q);
    n_new += is_new;
  }
  n_inst = wave_sum_u32(n_inst); n_hq = wave_sum_u32(n_hq);
  n_new = wave_sum_u32(n_new); n_foreign = wave_sum_u32(n_foreign);
  if (lane == 0) {
    if (count_instances) {
      if (n_inst) atomicAdd(reinterpret_cast<unsigned long long*>(&meta->instances), (unsigned long long)n_inst);
      if (n_hq) atomicAdd(reinterpret_cast<unsigned long long*>(&meta->hq_instances), (unsigned long long)n_hq);
      if (n_foreign) atomicAdd(reinterpret_cast<unsigned long long*>(&meta->foreign), (unsigned long long)n_foreign);
    }
    if (n_new) atomicAdd(reinterpret_cast<unsigned long long*>(&meta->new_keys), (unsigned long long)n_new);
  }
}

static inline unsigned grid_for(uint64_t n, unsigned block, unsigned max_blocks) {
  uint64_t b = (n + block - 1) / block;
  if (b < 1) b = 1;
  return (unsigned)(b > max_blocks ? max_blocks : b);
}

void qh_launch_insert_items(const uint64_t* items, uint64_t n, const unsigned long long* n_dev,
                            int count_instances, int hashed, qh_meta* meta, const qh_geom& g, uint64_t* table,
                            uint64_t* ovf, uint64_t ovf_cap, hipStream_t s) {
  hipLaunchKernelGGL(k_insert_items, dim3(grid_for(n, QH_BLOCK, 4096)), dim3(QH_BLOCK), 0, s, items, n, n_dev, meta,
                     g, table, ovf, ovf_cap, count_instances, hashed);
}

// old slot -> new table; the value field is carried over (mer_database.hpp:170-176)
__global__ __launch_bounds__(QH_BLOCK) void k_rehash(const uint64_t* __restrict__ old_table, const qh_geom og,
                                                     uint64_t* __restrict__ new_table, const qh_geom ng,
                                                     qh_meta* __restrict__ meta) {
  const uint64_t n = qh_table_slots(og);
  const uint64_t wmask = qh_low_mask(ng.wbits);
  for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t s = old_table[idx];
    if (!s) continue;
    const uint64_t val = qh_slot_val(og, s, idx);
    const qh_pos p = qh_locate_hash(ng, qh_slot_hash(og, s, idx));
    const uint64_t base = p.home & ~wmask;
    bool done = false;
    for (uint32_t i = 0; i <= (uint32_t)ng.max_reprobe && !done; ++i) {
      const uint64_t nidx = base | ((p.home + i) & wmask);
      unsigned long long* slot = reinterpret_cast<unsigned long long*>(new_table + nidx);
      // (a doubled table has one remainder bit fewer: a wide table may grow into the one-word form)
      const uint64_t nw = ((p.tag0 | (i + 1)) << ng.vbits) | (ng.wide ? 0ull : val);
      done = atomicCAS(slot, 0ull, (unsigned long long)nw) == 0;  // keys of the old table are distinct
      if (done && ng.wide) ng.wvals[nidx] = val;
    }
    if (!done) meta->rehash_fail = 1;
  }
}

void qh_launch_rehash(const uint64_t* old_table, const qh_geom& og, uint64_t* new_table, const qh_geom& ng,
                      qh_meta* meta, hipStream_t s) {
  hipLaunchKernelGGL(k_rehash, dim3(grid_for(qh_table_slots(og), QH_BLOCK, 8192)), dim3(QH_BLOCK), 0, s, old_table,
                     og, new_table, ng, meta);
}

// distinct (mer, value) pairs -> empty-or-disjoint table (reader of the jellyfish-layout file, qh_query_open)
__global__ __launch_bounds__(QH_BLOCK) void k_insert_valued(const uint64_t* __restrict__ mers,
                                                            const uint64_t* __restrict__ vals, uint64_t n,
                                                            uint64_t* __restrict__ table, const qh_geom g,
                                                            qh_meta* __restrict__ meta) {
  const uint64_t wmask = qh_low_mask(g.wbits);
  for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (uint64_t)gridDim.x * blockDim.x) {
    const qh_pos p = qh_locate(g, mers[idx] & g.kmask);
    const uint64_t val = vals[idx] & qh_low_mask(g.bits + 1);
    const uint64_t base = p.home & ~wmask;
    bool done = false;
    for (uint32_t i = 0; i <= (uint32_t)g.max_reprobe && !done; ++i) {
      const uint64_t nidx = base | ((p.home + i) & wmask);
      unsigned long long* slot = reinterpret_cast<unsigned long long*>(table + nidx);
      done = atomicCAS(slot, 0ull, (unsigned long long)(((p.tag0 | (i + 1)) << g.vbits) | (g.wide ? 0ull : val))) == 0;
      if (done && g.wide) g.wvals[nidx] = val;
    }
    if (!done) meta->rehash_fail = 1;
  }
}

void qh_launch_insert_valued(const uint64_t* mers, const uint64_t* vals, uint64_t n, uint64_t* table, const qh_geom& g,
                             qh_meta* meta, hipStream_t s) {
  hipLaunchKernelGGL(k_insert_valued, dim3(grid_for(n, QH_BLOCK, 8192)), dim3(QH_BLOCK), 0, s, mers, vals, n, table, g, meta);
}

// ---------------------------------------------------------------------- read side

// database_query::operator[] (mer_database.hpp:284-293). Probing is linear inside a window, so the probes of a key are
// consecutive words: the 64-byte line of eight slots that holds the home slot is loaded at once (HBM delivers a line
// for one slot anyway: the random 8-byte and 64-byte load rates are the same 55 G/s) and searched in registers; one
// lookup in six goes on to the next line at load 0.54. Slot by slot, every further probe is another dependent load.
struct qh_line { uint64_t s[8]; };
__device__ __forceinline__ uint64_t qh_lookup(const qh_geom& g, const uint64_t* __restrict__ table, uint64_t key) {
  const qh_pos p = qh_locate(g, key);
  if (p.shard != (uint32_t)g.shard_id) return 0;
  if (g.wbits >= 3) {
    const uint32_t wm = (1u << g.wbits) - 1u;
    uint64_t gaddr = p.home & ~7ull;          // first slot of the line
    int pi = -(int)((uint32_t)p.home & 7u);   // probe index of that slot (negative: before the home slot)
    uint64_t v = 0;
    bool pend = true;
    do {
      const qh_line L = *reinterpret_cast<const qh_line*>(table + gaddr);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = pi + j;
        const uint64_t sj = L.s[j];
        if (pend && idx >= 0) {
          if (idx > g.max_reprobe || sj == 0) pend = false;
          else if ((sj >> g.vbits) == (p.tag0 | (uint64_t)(idx + 1))) { v = qh_slot_val(g, sj, gaddr + j); pend = false; }
        }
      }
      gaddr = (gaddr & ~(uint64_t)wm) | (((uint32_t)gaddr + 8u) & wm);  // wraps at the window's end
      pi += 8;
    } while (pend);
    return v;
  }
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

__global__ __launch_bounds__(QH_BLOCK) void k_lookup(const uint64_t* __restrict__ table, const qh_geom g,
                                                     const uint64_t* __restrict__ mers, uint64_t n,
                                                     uint64_t* __restrict__ out, int get_val) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t m = mers[i];
    uint64_t v = (m & ~g.kmask) ? 0 : qh_lookup(g, table, m);
    if (get_val) v = (v & 1) ? v >> 1 : 0;  // mer_database.hpp:296-299
    out[i] = v;
  }
}

void qh_launch_lookup(const uint64_t* table, const qh_geom& g, const uint64_t* mers, uint64_t n, uint64_t* out,
                      int get_val, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_lookup, dim3(grid_for(n, QH_BLOCK, 8192)), dim3(QH_BLOCK), 0, s, table, g, mers, n, out, get_val);
}

// database lookup and contaminant membership of the same canonical mer in one pass
// (error_correct_reads.cc:348,390,620: is_contaminant(mer.canonical()) next to the
// database_query lookup). flags[i] = 1 iff the mer is in the contaminant set.
__global__ __launch_bounds__(QH_BLOCK) void k_lookup_contaminant(const uint64_t* __restrict__ table, const qh_geom g,
                                                                 const uint64_t* __restrict__ ctable, const qh_geom cg,
                                                                 const uint64_t* __restrict__ mers, uint64_t n,
                                                                 uint64_t* __restrict__ out,
                                                                 uint8_t* __restrict__ flags) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t m = mers[i];
    const bool ok = !(m & ~g.kmask);
    if (out) out[i] = ok ? qh_lookup(g, table, m) : 0;
    flags[i] = ok && qh_lookup(cg, ctable, m) != 0;
  }
}

void qh_launch_lookup_contaminant(const uint64_t* table, const qh_geom& g, const uint64_t* ctable, const qh_geom& cg,
                                  const uint64_t* mers, uint64_t n, uint64_t* out, uint8_t* flags, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_lookup_contaminant, dim3(grid_for(n, QH_BLOCK, 8192)), dim3(QH_BLOCK), 0, s, table, g, ctable, cg,
                     mers, n, out, flags);
}

// get_best_alternatives (mer_database.hpp:302-329): 4 lanes per mer, one per
// alternative base; lane 0 of each quad folds the four (count, hq) pairs in
// code order exactly like the upstream loop.
__global__ __launch_bounds__(QH_BLOCK) void k_best_alternatives(const uint64_t* __restrict__ table, const qh_geom g,
                                                                const uint64_t* __restrict__ fmers,
                                                                const uint8_t* __restrict__ backward, uint64_t n,
                                                                uint64_t* __restrict__ counts,
                                                                int32_t* __restrict__ ucode,
                                                                int32_t* __restrict__ level,
                                                                int32_t* __restrict__ nkept) {
  const uint64_t total = (n * 4 + blockDim.x - 1) / blockDim.x * blockDim.x;  // keep quads whole
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t i = t >> 2;
    const uint32_t alt = (uint32_t)(t & 3);
    uint64_t v = 0;
    if (i < n) {
      const unsigned sh = backward[i] ? 2u * (g.k - 1) : 0u;  // kmer.hpp:47-50,101-103
      const uint64_t m = ((fmers[i] & g.kmask) & ~(3ull << sh)) | ((uint64_t)alt << sh);
      v = qh_lookup(g, table, qh_canonical(m, g.k));
    }
    const int lane = threadIdx.x & 63, q0 = lane & ~3;
    uint64_t vv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) vv[a] = __shfl(v, q0 + a, 64);
    if (alt == 0 && i < n) {
      uint64_t c[4] = {0, 0, 0, 0};
      int cnt = 0, lvl = 0, uc = 0;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const uint64_t first = vv[a] >> 1;
        const int second = (int)(vv[a] & 1);
        if (first > 0 && second >= lvl) {
          if (second > lvl && cnt > 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (j < a) c[j] = 0;
            cnt = 0;
          }
          c[a] = first; uc = a; lvl = second; ++cnt;
        }
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) counts[i * 4 + a] = c[a];
      ucode[i] = uc; level[i] = lvl; nkept[i] = cnt;
    }
  }
}

// tables whose home slot may depend on the end bases (tiny k or tiny tables): four independent lookups
void qh_launch_best_alternatives_generic(const uint64_t* table, const qh_geom& g, const uint64_t* fmers,
                                         const uint8_t* backward, uint64_t n, uint64_t* counts, int32_t* ucode,
                                         int32_t* level, int32_t* nkept, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_best_alternatives, dim3(grid_for(n * 4, QH_BLOCK, 8192)), dim3(QH_BLOCK), 0, s, table, g,
                     fmers, backward, n, counts, ucode, level, nkept);
}

// error_correct_reads.cc:650-659: distinct += (v&1)&&(v>=2); total += v>>1
__global__ __launch_bounds__(QH_BLOCK) void k_stats(const uint64_t* __restrict__ table, const qh_geom g,
                                                    qh_meta* __restrict__ meta) {
  const uint64_t n = qh_table_slots(g);
  unsigned long long d = 0, t = 0, occ = 0;
  for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t s = table[idx];
    const uint64_t v = s ? qh_slot_val(g, s, idx) : 0;
    occ += s != 0;
    if ((v & 1) && v >= 2) { d += 1; t += v >> 1; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { d += __shfl_xor(d, o, 64); t += __shfl_xor(t, o, 64); occ += __shfl_xor(occ, o, 64); }
  if ((threadIdx.x & 63) == 0 && occ) {
    if (d) {
      atomicAdd(reinterpret_cast<unsigned long long*>(&meta->stat_distinct), d);
      atomicAdd(reinterpret_cast<unsigned long long*>(&meta->stat_total), t);
    }
    atomicAdd(reinterpret_cast<unsigned long long*>(&meta->stat_occupied), occ);  // (the table's load: the corrector sizes its copy by it)
  }
}

void qh_launch_stats(const uint64_t* table, const qh_geom& g, qh_meta* meta, hipStream_t s) {
  hipLaunchKernelGGL(k_stats, dim3(grid_for(qh_table_slots(g), QH_BLOCK, 4096)), dim3(QH_BLOCK), 0, s, table, g, meta);
}

// histo_mer_database.cc:13-21
__global__ __launch_bounds__(QH_BLOCK) void k_histo(const uint64_t* __restrict__ table, const qh_geom g,
                                                    unsigned long long* __restrict__ histo) {
  __shared__ uint32_t h[1001 * 2];
  for (int i = threadIdx.x; i < 1001 * 2; i += blockDim.x) h[i] = 0;
  __syncthreads();
  const uint64_t n = qh_table_slots(g);
  for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t s = table[idx];
    if (!s) continue;
    const uint64_t v = qh_slot_val(g, s, idx);
    uint64_t c = v >> 1;
    if (c > 1000) c = 1000;
    atomicAdd(&h[c * 2 + (v & 1)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 1001 * 2; i += blockDim.x)
    if (h[i]) atomicAdd(&histo[i], (unsigned long long)h[i]);
}

void qh_launch_histo(const uint64_t* table, const qh_geom& g, unsigned long long* histo, hipStream_t s) {
  // each block's 32-bit LDS bins see at most slots/blocks entries
  hipLaunchKernelGGL(k_histo, dim3(grid_for(qh_table_slots(g), QH_BLOCK, 2048)), dim3(QH_BLOCK), 0, s, table, g, histo);
}

// database_query iteration (mer_database.hpp:331-361) as a compacting export
__global__ __launch_bounds__(QH_BLOCK) void k_dump(const uint64_t* __restrict__ table, const qh_geom g,
                                                   uint64_t* __restrict__ mers, uint64_t* __restrict__ vals,
                                                   uint64_t cap, qh_meta* __restrict__ meta) {
  const uint64_t n = qh_table_slots(g);
  const int lane = threadIdx.x & 63;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t rounds = (n + stride - 1) / stride;
  for (uint64_t rd = 0; rd < rounds; ++rd) {
    const uint64_t idx = rd * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t s = idx < n ? table[idx] : 0;
    const uint64_t mask = __ballot(s != 0);
    if (!mask) continue;
    unsigned long long base = 0;
    const int leader = __ffsll((unsigned long long)mask) - 1;
    if (lane == leader)
      base = atomicAdd(reinterpret_cast<unsigned long long*>(&meta->dump_count), (unsigned long long)__popcll(mask));
    base = __shfl(base, leader, 64);
    if (s) {
      const unsigned long long at = base + lanes_below(mask);
      if (at < cap) { mers[at] = qh_slot_key(g, s, idx); vals[at] = qh_slot_val(g, s, idx); }
    }
  }
}

void qh_launch_dump(const uint64_t* table, const qh_geom& g, uint64_t* mers, uint64_t* vals, uint64_t cap,
                    qh_meta* meta, hipStream_t s) {
  hipLaunchKernelGGL(k_dump, dim3(grid_for(qh_table_slots(g), QH_BLOCK, 4096)), dim3(QH_BLOCK), 0, s, table, g, mers,
                     vals, cap, meta);
}

// --------------------------------------------------------- roofline microbenchmarks

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// The values image of the database file (atomic_bits_array layout: 64 / vbits value fields per
// 64-bit word, none straddling a word; mer_database.hpp:115-126 writes vals_ raw): word wi of the
// whole-file image for this shard's slots [first, first + slots).
__global__ __launch_bounds__(QH_BLOCK) void k_vals_image(const uint64_t* __restrict__ table, const qh_geom g,
                                                         uint64_t first, uint64_t slots, uint64_t w0, uint64_t nwords,
                                                         uint64_t* __restrict__ out) {
  const uint32_t vb = (uint32_t)g.bits + 1u, vpw = 64u / vb;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t base = (w0 + i) * vpw;
    uint64_t word = 0;
    for (uint32_t j = 0; j < vpw; ++j) {
      const uint64_t gi = base + j;
      if (gi >= first && gi - first < slots) {
        const uint64_t s = table[gi - first];
        if (s) word |= qh_slot_val(g, s, gi - first) << (j * vb);
      }
    }
    out[i] = word;
  }
}

// the reverse for a table in the wide form read from a file: value fields of the image -> side array
__global__ __launch_bounds__(QH_BLOCK) void k_vals_unpack(const uint64_t* __restrict__ image, int vb, uint64_t slots,
                                                          uint64_t* __restrict__ wvals) {
  const uint32_t vpw = 64u / (uint32_t)vb;
  const uint64_t vm = qh_low_mask(vb);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (uint64_t)gridDim.x * blockDim.x)
    wvals[i] = (image[i / vpw] >> ((i % vpw) * (uint32_t)vb)) & vm;
}

void qh_launch_vals_unpack(const uint64_t* image, int value_bits, uint64_t slots, uint64_t* wvals, hipStream_t s) {
  if (!slots) return;
  hipLaunchKernelGGL(k_vals_unpack, dim3(grid_for(slots, QH_BLOCK, 16384)), dim3(QH_BLOCK), 0, s, image, value_bits, slots, wvals);
}

void qh_launch_vals_image(const uint64_t* table, const qh_geom& g, uint64_t first, uint64_t slots, uint64_t w0,
                          uint64_t nwords, uint64_t* out, hipStream_t s) {
  if (!nwords) return;
  hipLaunchKernelGGL(k_vals_image, dim3(grid_for(nwords, QH_BLOCK, 16384)), dim3(QH_BLOCK), 0, s, table, g, first, slots,
                     w0, nwords, out);
}

// mode 0: one 64-bit atomicCAS per update at a uniformly random slot
// mode 1: one 8-byte load per update
// mode 2: relaxed load followed by a dependent CAS (the insert path's shape)
__global__ __launch_bounds__(QH_BLOCK) void k_random_access(uint64_t* __restrict__ table, int lsize, uint64_t n,
                                                            int mode, uint64_t seed,
                                                            unsigned long long* __restrict__ sink) {
  const uint64_t mask = (1ull << lsize) - 1;
  unsigned long long acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = mix64(seed + i * 0x9e3779b97f4a7c15ull);
    unsigned long long* p = reinterpret_cast<unsigned long long*>(table + (r & mask));
    if (mode == 0) {
      acc += atomicCAS(p, 0ull, (unsigned long long)(r | 1));
    } else if (mode == 1) {
      acc += *p;
    } else {
      const unsigned long long old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc += atomicCAS(p, old, old + 2);
    }
  }
  if (acc == 0x123456789abcdefull) *sink = acc;  // keep the loads alive
}

void qh_launch_random_access(uint64_t* table, int lsize, uint64_t n, int mode, uint64_t seed,
                             unsigned long long* sink, hipStream_t s) {
  hipLaunchKernelGGL(k_random_access, dim3(8192), dim3(QH_BLOCK), 0, s, table, lsize, n, mode, seed, sink);
}

// LDS operation-rate microbenchmark (design input for k_window_build): every lane performs `iters`
// operations at pseudo-random slots of a 128 KiB LDS array.
// mode 0 ds_read_b64, 1 ds_write_b64, 2 ds_cmpst_rtn_b64, 3 ds_cmpst_rtn_b32, 4 ds_add_rtn_u32,
//      5 ds_add_u32 (no return), 6 ds_add_rtn_u64, 7 read_b64 + write_b64 + read_b64 (write-verify)
__global__ __launch_bounds__(1024) void k_lds_rate(int mode, int iters, unsigned long long* __restrict__ sink) {
  __shared__ unsigned long long a[16384];
  unsigned int* a32 = reinterpret_cast<unsigned int*>(a);
  for (int i = threadIdx.x; i < 16384; i += 1024) a[i] = i;
  __syncthreads();
  unsigned int x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  unsigned long long acc = 0;
  for (int i = 0; i < iters; ++i) {
    x = x * 1664525u + 1013904223u;
    const unsigned int idx = (x >> 10) & 16383u;
    switch (mode) {
      case 0: acc += __hip_atomic_load(&a[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break;
      case 1: __hip_atomic_store(&a[idx], (unsigned long long)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break;
      case 2: acc += atomicCAS(&a[idx], (unsigned long long)idx, (unsigned long long)x); break;
      case 3: acc += atomicCAS(&a32[idx * 2], idx, x); break;
      case 4: acc += atomicAdd(&a32[idx * 2], 1u); break;
      case 5: atomicAdd(&a32[idx * 2], 1u); break;
      case 6: acc += atomicAdd(&a[idx], 1ull); break;
      default: {
        const unsigned long long o = __hip_atomic_load(&a[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&a[idx], o + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        acc += __hip_atomic_load(&a[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  if (acc == 0x123456789abcdefull) *sink = acc;
}

void qh_launch_lds_rate(int mode, int iters, int blocks, unsigned long long* sink, hipStream_t s) {
  hipLaunchKernelGGL(k_lds_rate, dim3(blocks), dim3(1024), 0, s, mode, iters, sink);
}
