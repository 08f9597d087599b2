// qh_partition.hip -- the radix-partitioned build: the same hash_with_quality::add
// (src/mer_database.hpp:94-113) over the same table as qh_kernels.hip's direct path,
// restructured so that no k-mer update is a global atomic.
//
// Why: a device-scope 64-bit atomicCAS on MI355X executes at the memory side at
// ~22 G/s chip-wide whatever the footprint (measured, profiles/r01_v1_direct), and
// each one moves a 64-B line in and out of HBM for 8 useful bytes. Here the items
// of a chunk are first bucketed by the 16384-slot probe WINDOW their key hashes to
// (coalesced streaming writes staged through LDS, one global atomic per
// (tile, bucket)), then one workgroup per window pulls the window's 128 KiB into
// LDS, applies its bucket with LDS compare-and-swap, and streams the window back.
// HBM sees only streaming traffic: 8 B/item written + 8 B/item read per level plus
// one read+write of the table.
//
//   k_rec_blocks_*     record table -> exclusive prefix of 64-base blocks per record
//   k_partition<SRC>   FASTQ records (fused k-mer extraction, create_database.cc:64-92)
//                      or an item list -> buckets by a bit field of the hash
//   k_window_build     bucket -> LDS window -> table
#include "qh_kernels.h"
#include "qh_device.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#define QP_THREADS 1024
#define QP_WAVES (QP_THREADS / 64)
#define QP_BLOCKS_PER_WAVE 12
#define QP_TILE_BLOCKS (QP_WAVES * QP_BLOCKS_PER_WAVE)  // 192 blocks of 64 positions
#define QP_TILE_ITEMS (QP_TILE_BLOCKS * 64)             // 12288 items = 96 KiB of LDS
#define QP_MAX_BINS 4096
#define QP_NONE (~0ull)

// ------------------------------- bit planes and segment descriptors of a FASTQ chunk
//
// TEXT PLANES: four bit arrays with one bit per BYTE of the chunk, whatever the byte is part of:
//   v  the byte is A/C/G/T in either case      l0, l1  its 2-bit code (A 0, C 1, G 2, T 3)
//   h  the byte, read as a signed char quality, is >= the threshold (create_database.cc:81)
// stored as 64-bit words (word j = bytes 64j .. 64j+63), plane after plane. A read's bases are
// the v/l0/l1 bits at its sequence offset and the h bits at its quality offset, so the planes are
// made by one streaming pass that knows nothing about records.
//
// SEGMENTS: up to QP_SEG consecutive k-mers of one read, the unit of work of one extraction lane:
// {byte offset of the first base, (distance to its quality byte) << 4 | k-mers}; all zero = no segment. A read's
// k-mers are spread evenly over its segments. The count matters: a window past it is a k-mer of the next
// segment, or runs over the newline after the read (no base) or, for k < QP_SEG, can even lie wholly behind
// that newline, inside the quality string, whose letters may all be A, C, G, T.

#define RB_TILE 1024  // records per scan tile
#define QP_SEG 12     // k-mers per segment (= items a lane keeps in registers per tile)
static_assert(QP_SEG == QH_SEG_KMERS && QP_SEG == QP_BLOCKS_PER_WAVE, "segment size is shared with the host-side buffer sizing");

__device__ __forceinline__ uint32_t rec_nsegs(uint32_t len, uint32_t k) {
  return len >= k ? (len - k + 1 + QP_SEG - 1) / QP_SEG : 0;
}

// tile_sums: [0, ntiles) segments per tile of RB_TILE records; the scan leaves the total at [ntiles]
// Also clears the bucket counters of the partition levels (zero_a, zero_b: 16-byte pieces): two fewer fill
// launches of ~18 us each in front of the first partition kernel.
__global__ __launch_bounds__(256) void k_rec_segs_sum(const uint4* __restrict__ recs, const qh_meta* __restrict__ meta,
                                                      uint32_t* __restrict__ tile_sums, uint32_t k,
                                                      uint4* __restrict__ zero_a, uint32_t na,
                                                      uint4* __restrict__ zero_b, uint32_t nb) {
  __shared__ uint32_t ws[4];
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < na; i += gridDim.x * 256) zero_a[i] = make_uint4(0, 0, 0, 0);
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nb; i += gridDim.x * 256) zero_b[i] = make_uint4(0, 0, 0, 0);
  const uint32_t nrec = meta->format_err ? 0 : meta->nrec;
  const uint32_t base = blockIdx.x * RB_TILE;
  uint32_t t = 0;
  for (uint32_t r = base + threadIdx.x; r < base + RB_TILE && r < nrec; r += 256) t += rec_nsegs(recs[r].y, k);
  t = wave_sum_u32(t);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// exclusive scan of the segment counts of n tiles of records in place; the total goes to v[n] (block sums, their scan,
// apply: qh_device.h). n comes from the record table's capacity (a record per 16 bytes of text); only the tiles that
// hold records are scanned (the others hold zero and nobody reads their prefix): a twelfth of them for 100-base reads
__global__ __launch_bounds__(256) void k_scan_u32_sums(const uint32_t* __restrict__ v, uint32_t n, uint32_t* __restrict__ sums,
                                                       const qh_meta* __restrict__ meta) {
  qs_block_sums(v, qs_n_eff(n, &meta->nrec, RB_TILE), sums);
}
__global__ __launch_bounds__(1024) void k_scan_u32_top(uint32_t* __restrict__ v, uint32_t n, uint32_t* __restrict__ sums, uint32_t nb) {
  const uint32_t total = qs_top(sums, nb);
  if (threadIdx.x == 1023) v[n] = total;
}
__global__ __launch_bounds__(256) void k_scan_u32_apply(uint32_t* __restrict__ v, uint32_t n, const uint32_t* __restrict__ sums,
                                                        const qh_meta* __restrict__ meta) {
  qs_apply(v, qs_n_eff(n, &meta->nrec, RB_TILE), sums);
}

__global__ __launch_bounds__(256) void k_rec_segs_apply(const uint4* __restrict__ recs, qh_meta* __restrict__ meta,
                                                        const uint32_t* __restrict__ tile_sums, uint32_t k,
                                                        uint2* __restrict__ seg_desc) {
  // 128 reads a round (half the workgroup computes, all of it copies out): 128 reads of 100 to 150 bases make 900 to
  // 1300 segments, and the 12 KiB stage fits beside another chunk's level-1 kernel (147.5 of the CU's 160 KiB)
  constexpr uint32_t SEG_STAGE = 1536, SEG_ROUND = 128;
  __shared__ uint32_t wtot[4];
  __shared__ uint2 stage[SEG_STAGE];
  const uint32_t nrec = meta->format_err ? 0 : meta->nrec;
  const uint32_t base = blockIdx.x * RB_TILE;
  if (base >= nrec) return;
  uint32_t srun = tile_sums[blockIdx.x];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (uint32_t j = 0; j < RB_TILE; j += SEG_ROUND) {
    const uint32_t r = base + j + threadIdx.x;
    const uint4 rc = threadIdx.x < SEG_ROUND && r < nrec ? recs[r] : make_uint4(0, 0, 0, 0);
    const uint32_t cs = rec_nsegs(rc.y, k);
    uint32_t sinc = cs;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(sinc, o, 64);
      if (lane >= o) sinc += y;
    }
    if (lane == 63) wtot[wv] = sinc;
    __syncthreads();
    uint32_t swbase = 0, stot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t y = wtot[w];
      if (w < wv) swbase += y;
      stot += y;
    }
    // the descriptors of these 256 records are consecutive: staged in LDS and written as one run (a thread's own
    // stores lie 8 * cs bytes apart from its neighbour's); reads of thousands of bases write where they belong
    const bool staged = stot <= SEG_STAGE;
    if (cs) {
      // k-mers [a, a + n) of the read; k-mer i spans bases i .. i+k-1
      const uint32_t ps = swbase + sinc - cs, nk = rc.y - k + 1;
      uint32_t dist = rc.z - rc.x;  // bases, newline, separator line, newline
      if (dist >> 28) {             // a line of 256 MiB
        atomicOr(&meta->format_err, QH_FMT_LINE_TOO_LONG);
        atomicMin(&meta->first_bad, r);
        dist = 0;
      }
      // the read's k-mers in cs segments of equal size (+1 for the first few): the lanes of a wave then run the
      // same number of windows (77 k-mers are 7 x 11, not 6 x 12 + 5)
      const uint32_t each = nk / cs, more = nk - each * cs;
      for (uint32_t b = 0; b < cs; ++b) {
        const uint32_t a = b * each + (b < more ? b : more);
        const uint2 d = make_uint2(rc.x + a, (dist << 4) | (each + (b < more ? 1u : 0u)));
        if (staged) stage[ps + b] = d;
        else seg_desc[srun + ps + b] = d;
      }
    }
    __syncthreads();
    if (staged)
      for (uint32_t i = threadIdx.x; i < stot; i += 256) seg_desc[srun + i] = stage[i];
    srun += stot;
    __syncthreads();
  }
}

// a lane per 16 bytes of text -> 16 bits of each plane; 1 KiB of text per wave and round
__global__ __launch_bounds__(256) void k_text_planes(const uint8_t* __restrict__ buf, size_t len,
                                                     uint16_t* __restrict__ planes, size_t plane_halfwords,
                                                     int min_qual) {
  const qh_swar_thr thr = qh_swar_threshold(min_qual);
  const size_t nunits = (len + 15) / 16;
  for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < nunits; u += (size_t)gridDim.x * 256)
    qh_store_text_planes(load16_guarded(buf, u * 16, len), u, planes, plane_halfwords, thr);
}

void qh_launch_text_planes(const uint8_t* buf, size_t len, const qh_geom& g, const qh_fastq_aux& a, hipStream_t s) {
  if (!len) return;
  const size_t nunits = (len + 15) / 16;
  size_t wgs = (nunits + 255) / 256;
  if (wgs > 256 * 8 * 4) wgs = 256 * 8 * 4;
  hipLaunchKernelGGL(k_text_planes, dim3((unsigned)wgs), dim3(256), 0, s, buf, len, reinterpret_cast<uint16_t*>(a.planes),
                     a.plane_words * 4, g.min_qual);
}

void qh_launch_segments(const uint4* recs, uint32_t rec_cap, qh_meta* meta, const qh_geom& g,
                        const qh_fastq_aux& a, void* zero_a, size_t zero_a_bytes, void* zero_b, size_t zero_b_bytes,
                        hipStream_t s) {
  const uint32_t ntiles = (rec_cap + RB_TILE - 1) / RB_TILE;
  hipLaunchKernelGGL(k_rec_segs_sum, dim3(ntiles), dim3(256), 0, s, recs, meta, a.tile_sums, (uint32_t)g.k,
                     static_cast<uint4*>(zero_a), (uint32_t)(zero_a_bytes / 16), static_cast<uint4*>(zero_b),
                     (uint32_t)(zero_b_bytes / 16));
  // (tile_sums has room for ntiles + 1 sums and QH_SCAN_SCRATCH words of scratch behind them)
  uint32_t* sums = a.tile_sums + ntiles + 1;
  const uint32_t nb = (ntiles + QS_BLOCK - 1) / QS_BLOCK;
  if (nb) hipLaunchKernelGGL(k_scan_u32_sums, dim3(nb), dim3(256), 0, s, a.tile_sums, ntiles, sums, meta);
  hipLaunchKernelGGL(k_scan_u32_top, dim3(1), dim3(1024), 0, s, a.tile_sums, ntiles, sums, nb);
  if (nb) hipLaunchKernelGGL(k_scan_u32_apply, dim3(nb), dim3(256), 0, s, a.tile_sums, ntiles, sums, meta);
  hipLaunchKernelGGL(k_rec_segs_apply, dim3(ntiles), dim3(256), 0, s, recs, meta, a.tile_sums, (uint32_t)g.k, a.seg_desc);
}

// ------------------------------------------------------------- LDS-staged multi-split

struct qp_params {
  // destination buckets: bucket ob holds out[ob * out_cap ...], its fill count in out_counts[ob]
  uint64_t* out;
  uint64_t out_cap;
  unsigned long long* out_counts;
  uint32_t out_count_stride;  // counters of consecutive buckets lie this many words apart (contention, see qh_kernels.h)
  int bin_shift;       // bin = (h >> bin_shift) & bin_mask
  uint32_t bin_mask;
  uint32_t bin_bits;
  int out_bin_shift;   // ob = (input bucket << out_bin_shift) | bin
  uint64_t* spill;     // items that do not fit their bucket (meta->spill_count), inserted directly later
  uint64_t spill_cap;
  qh_meta* meta;
  int check_owner;     // drop (and count) items whose owner shard is not g.shard_id
  int count_instances; // add emitted items to meta->instances / hq_instances
  // SRC_FASTQ: the chunk as text planes + segment descriptors (qh_launch_text_planes / _segments)
  const uint64_t* planes;
  uint64_t plane_words;  // words per plane
  const uint2* seg_desc;
  const uint32_t* seg_total;
  // SRC_ITEMS: input bucket b = blockIdx.y holds in[b * in_cap ...], in_counts[b] items (or in_n when in_counts == 0)
  const uint64_t* in;
  uint64_t in_cap;
  const unsigned long long* in_counts;
  uint32_t in_count_stride;
  uint64_t in_n;
  int in_hashed;       // 0: (mer<<1|q) as the C ABI passes them, 1: (hash<<1|q)
  int out_pack_gsize;  // > 0: the buckets take packed 32-bit items (qh_part_dst::pack_gsize)
  int out_pack_sbits;  // ... the table's shard_bits
};

enum { QP_SRC_FASTQ = 0, QP_SRC_ITEMS = 1 };

__device__ __forceinline__ void qp_spill(const qp_params& p, uint64_t item) {
  const unsigned long long at = atomicAdd(reinterpret_cast<unsigned long long*>(&p.meta->spill_count), 1ull);
  if (at < p.spill_cap) p.spill[at] = item;
  else p.meta->ovf_lost = 1;
}

// OWNER: items are checked against the shard that owns them (p.check_owner); KCLASS: 2 is k < 4 (no core in the
// hash; 0 and 1 are compiled alike). Both are the same for
// every item of a launch, and the extraction loop is bound by the instructions it issues per k-mer.
// KFIX: the mer length when it is known at compile time (24 = the quorum driver's default, 31 = the largest), else 0: shift counts
// and masks of the extraction loop fold into the instructions.
// BINLO: the bin field of (hash << 1 | q) lies in the low 32 bits (bin_shift + bin_bits <= 31): one bit-field extract
// instead of a 64-bit shift and a mask, three times per item.
// BPW, MAXB: 64-item blocks a wave keeps per tile and the most buckets a launch may split into. The tile (16 x BPW x 64
// items) and the three per-bucket arrays share the CU's LDS: 12 blocks with 4096 buckets by default (a lane of the FASTQ
// source keeps a segment of QP_SEG = 12 k-mers), 18 blocks with 1024 buckets (156 KiB, 128 VGPRs, nothing spilled) for
// item sources that split 512 or 1024 ways -- runs half as long again where they are shortest, level 2 of the large
// tables: 17.4 -> 16.3 ms per configs[2] step; 16 blocks 16.6 ms; no gain at 64 or 128 buckets.
template <int SRC, bool OWNER, int KCLASS, int KFIX = 0, bool BINLO = false, int BPW = QP_BLOCKS_PER_WAVE, int MAXB = QP_MAX_BINS>
__global__ __launch_bounds__(QP_THREADS) void k_partition(const qp_params p, const qh_geom g) {
  const int K = KFIX ? KFIX : g.k;
  auto bin_of_item = [&](uint64_t item) -> uint32_t {  // item = hash << 1 | q
    return BINLO ? __builtin_amdgcn_ubfe((uint32_t)item, (uint32_t)p.bin_shift + 1u, p.bin_bits)
                 : (uint32_t)((item >> 1) >> p.bin_shift) & p.bin_mask;
  };
  constexpr uint32_t TILE_ITEMS = QP_WAVES * BPW * 64;
  static_assert(SRC != QP_SRC_FASTQ || BPW == QP_SEG, "a lane of the FASTQ source keeps one segment");
  __shared__ uint64_t sorted[TILE_ITEMS];
  __shared__ uint32_t cursor[MAXB];  // histogram, then running placement cursor
  __shared__ uint32_t offs[MAXB];    // exclusive prefix of the histogram
  __shared__ uint32_t gbase[MAXB];   // reserved start inside the global bucket
  __shared__ uint32_t wsum[QP_WAVES];
  __shared__ uint32_t s_total;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t nbins = p.bin_mask + 1;
  const uint32_t smask = (uint32_t)qh_low_mask(g.shard_bits);
  uint32_t n_inst = 0, n_hq = 0, n_foreign = 0;

  uint64_t ntiles;
  uint32_t nseg = 0;
  const uint64_t* src = nullptr;
  uint64_t src_n = 0;
  if (SRC == QP_SRC_FASTQ) {
    if (p.meta->format_err) return;
    nseg = *p.seg_total;
    ntiles = ((uint64_t)nseg + QP_THREADS - 1) / QP_THREADS;
  } else {
    src = p.in + (uint64_t)blockIdx.y * p.in_cap;
    src_n = p.in_counts ? (uint64_t)p.in_counts[(size_t)blockIdx.y * p.in_count_stride] : p.in_n;
    if (src_n > p.in_cap) src_n = p.in_cap;
    ntiles = (src_n + TILE_ITEMS - 1) / TILE_ITEMS;
  }
  const uint32_t in_bucket = SRC == QP_SRC_ITEMS ? blockIdx.y : 0;

  // Software pipeline over this workgroup's tiles: the inputs of tile t+1 are requested while
  // tile t is sorted and written out (one workgroup per CU: nothing else would hide the latency).
  // FASTQ: a lane per segment; its descriptor is fetched two tiles ahead, its two plane blocks
  // (64 bytes) one tile ahead.
  const uint2 no_seg = make_uint2(0, 0);
  uint2 sd_cur = no_seg, sd_next = no_seg;
  uint4 pl[4];  // words j, j+1 of the v, l0, l1 planes at the sequence offset and of h at the quality offset
  uint64_t raw[BPW];
  auto load_sdesc = [&](uint64_t tile) -> uint2 {
    const uint64_t sidx = tile * QP_THREADS + tid;
    return sidx < nseg ? p.seg_desc[sidx] : no_seg;
  };
  auto load2 = [&](const uint64_t* q) -> uint4 {  // two consecutive words, 8-byte aligned
    const uint2 a = *reinterpret_cast<const uint2*>(q), b = *reinterpret_cast<const uint2*>(q + 1);
    return make_uint4(a.x, a.y, b.x, b.y);
  };
  auto issue_planes = [&](uint2 sd) {  // no segment: the chunk's first words
    const uint64_t* ps = p.planes + (sd.x >> 6);
    pl[0] = load2(ps);
    pl[1] = load2(ps + 2 * p.plane_words);
    pl[2] = load2(ps + 3 * p.plane_words);
    pl[3] = load2(p.planes + p.plane_words + ((sd.x + (sd.y >> 4)) >> 6));
  };
  auto issue_items = [&](uint64_t tile) {
    const uint64_t j0 = tile * TILE_ITEMS + (uint64_t)wave * (BPW * 64) + lane;
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
      const uint64_t j = j0 + (uint64_t)i * 64;
      raw[i] = j < src_n ? src[j] : QP_NONE;
    }
  };
  if (SRC == QP_SRC_FASTQ) {
    sd_cur = load_sdesc(blockIdx.x);
    issue_planes(sd_cur);
    sd_next = load_sdesc((uint64_t)blockIdx.x + gridDim.x);
  } else {
    issue_items(blockIdx.x);
  }

  for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (uint32_t b = tid; b < nbins; b += QP_THREADS) cursor[b] = 0;
    __syncthreads();

    // ---- pass 1: produce this wave's items (kept in registers) and the tile histogram
    // FASTQ source: bit i of live = it[i] holds an item of this shard; item source: QP_NONE marks the others
    // (measured: the mask costs level 2 4 %, the sentinel's initialisation costs level 1 its 12 moves)
    uint64_t it[BPW];
    uint32_t live = 0;
    if (SRC == QP_SRC_ITEMS) {
#pragma unroll
      for (int i = 0; i < BPW; ++i) it[i] = QP_NONE;
    }

    if (SRC == QP_SRC_FASTQ) {
      // create_database.cc:64-92 for the segment's k-mers: no rolling registers, every k-mer is a
      // k-bit window of the planes. w* hold bases a .. a+63 of the read (bit i = base a+i): v, l0, l1
      // from the sequence bytes, h from the quality bytes.
      const uint32_t o = sd_cur.x & 63u, oq = (sd_cur.x + (sd_cur.y >> 4)) & 63u, cnt = sd_cur.y & 15u;
      if (cnt) {
        // (k = 31: the mask comes from the geometry; as a literal, 2^31 - 1 crashes the instruction selection of clang 22)
        const uint32_t kones = KFIX == 31 ? (uint32_t)((1ull << g.k) - 1) : (uint32_t)((1ull << K) - 1);
        auto fun = [&](const uint4& w2, uint32_t sh) -> uint64_t {
          const uint64_t lo = ((uint64_t)w2.y << 32) | w2.x, hi = ((uint64_t)w2.w << 32) | w2.z;
          return (lo >> sh) | ((hi << 1) << (63 - sh));
        };
        // bases past the segment's last k-mer count as no bases: the windows past its count are then invalid
        const uint64_t wv = fun(pl[0], o) & ((1ull << (cnt + K - 1)) - 1), w0 = fun(pl[1], o), w1 = fun(pl[2], o), wh = fun(pl[3], oq);
        // low_len >= k (:85) and high_len >= k (:86) for the segment's windows at once: bit j of runs(x) = bits
        // j .. j+K-1 of x are all ones (log-step ANDs), instead of a mask-and-compare per window and plane
        auto runs = [&](uint64_t x) -> uint32_t {
          int have = 1;
          while (have * 2 <= K) { x &= x >> have; have *= 2; }
          if (have < K) x &= x >> (K - have);
          return (uint32_t)x & ((1u << QP_SEG) - 1u);
        };
        const uint32_t vm = runs(wv), hm = runs(wh) & vm;
        if (!OWNER) { n_inst += __popc(vm); n_hq += __popc(hm); live = vm; }
        // forward orientation: bit-reversed planes, aligned so that the newest base of k-mer j sits
        // at bit QP_SEG-1-j
        const int sh0 = 64 - K - (QP_SEG - 1);
        const uint64_t f0 = __builtin_bitreverse64(w0) >> sh0, f1 = __builtin_bitreverse64(w1) >> sh0;
#pragma unroll
        for (int j = 0; j < QP_SEG; ++j) {
          if ((vm >> j) & 1u) {
            const uint32_t q = (hm >> j) & 1u;
            const uint32_t r0 = ~(uint32_t)(w0 >> j) & kones, r1 = ~(uint32_t)(w1 >> j) & kones;
            const uint32_t m0 = (uint32_t)(f0 >> (QP_SEG - 1 - j)) & kones, m1 = (uint32_t)(f1 >> (QP_SEG - 1 - j)) & kones;
            const uint64_t h = qh_hash_strands_form<KCLASS == 2>(m0, m1, r0, r1, K);  // min(m, rm) of create_database.cc:86, hashed
            if (OWNER && ((uint32_t)h & smask) != (uint32_t)g.shard_id) {
              ++n_foreign;
            } else {
              if (OWNER) { ++n_inst; n_hq += q; live |= 1u << j; }
              it[j] = (h << 1) | q;
              atomicAdd(&cursor[bin_of_item(it[j])], 1u);
            }
          }
        }
      }
      // next tile: its planes now (descriptors arrived an iteration ago), the descriptors after it
      sd_cur = sd_next;
      issue_planes(sd_cur);
      sd_next = load_sdesc(tile + 2ull * gridDim.x);
    } else {
#pragma unroll
      for (int i = 0; i < BPW; ++i) {
        if (raw[i] != QP_NONE) {
          const uint32_t q = (uint32_t)(raw[i] & 1);
          const uint64_t h = p.in_hashed ? ((raw[i] >> 1) & g.hmask) : qh_hash_key((raw[i] >> 1) & g.kmask, g.k);
          if (OWNER && ((uint32_t)h & smask) != (uint32_t)g.shard_id) {
            ++n_foreign;
          } else {
            ++n_inst; n_hq += q;
            it[i] = (h << 1) | q;
            atomicAdd(&cursor[bin_of_item(it[i])], 1u);
          }
        }
      }
      issue_items(tile + gridDim.x);
    }
    __syncthreads();

    // ---- exclusive scan of the histogram (4 bins per thread), bucket space reservation
    uint32_t resv[4] = {0, 0, 0, 0};
    {
      uint32_t c[4], s = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t b = tid * 4 + e;
        c[e] = b < nbins ? cursor[b] : 0;
        s += c[e];
      }
      uint32_t inc = s;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t x = __shfl_up(inc, o, 64);
        if (lane >= o) inc += x;
      }
      if (lane == 63) wsum[wave] = inc;
      __syncthreads();
      uint32_t wbase = 0, tot = 0;
#pragma unroll
      for (int w = 0; w < QP_WAVES; ++w) {
        const uint32_t x = wsum[w];
        if (w < wave) wbase += x;
        tot += x;
      }
      if (tid == 0) s_total = tot;
      uint32_t run = wbase + inc - s;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t b = tid * 4 + e;
        if (b < nbins) {
          offs[b] = run;
          cursor[b] = run;
          // reserve the bucket range now, look at the answer only after the placement pass
          resv[e] = c[e] ? (uint32_t)atomicAdd(&p.out_counts[(((uint64_t)in_bucket << p.out_bin_shift) | b) * p.out_count_stride],
                                               (unsigned long long)c[e])
                         : 0u;
          run += c[e];
        }
      }
    }
    __syncthreads();

    // ---- pass 2: place the items bucket-sorted into LDS
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
      if (SRC == QP_SRC_ITEMS ? it[i] != QP_NONE : ((live >> i) & 1u) != 0) {
        sorted[atomicAdd(&cursor[bin_of_item(it[i])], 1u)] = it[i];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if ((uint32_t)tid * 4 + e < nbins) gbase[tid * 4 + e] = resv[e];
    __syncthreads();

    // ---- stream the tile out: consecutive lanes of one bucket write consecutive words
    const uint32_t total = s_total;
    for (uint32_t j = tid; j < total; j += QP_THREADS) {
      const uint64_t v = sorted[j];
      const uint32_t b = bin_of_item(v);
      const uint64_t dst = (uint64_t)gbase[b] + (j - offs[b]);
      if (dst < p.out_cap) {
        const uint64_t oidx = (((uint64_t)in_bucket << p.out_bin_shift) | b) * p.out_cap + dst;
        if (p.out_pack_gsize) {  // window buckets of a table whose remainder is short: 4 bytes per item
          const uint64_t hh = v >> 1;
          reinterpret_cast<uint32_t*>(p.out)[oidx] =
              ((uint32_t)v & 1u) | (((uint32_t)(hh >> p.out_pack_sbits) & ((1u << QH_WINDOW_BITS) - 1u)) << 1) |
              ((uint32_t)(hh >> p.out_pack_gsize) << (QH_WINDOW_BITS + 1));
        } else {
          uint64_t* at = &p.out[oidx];
          // level 1's buckets are read once, by level 2, after all of them are written: kept out of the caches
          // (level 2 then runs 5 % faster; the same hint on level 2's own stores or on the item loads costs time)
          if (SRC == QP_SRC_FASTQ) __builtin_nontemporal_store(v, at);
          else *at = v;
        }
      }
      else qp_spill(p, v);
    }
    __syncthreads();
  }

  if (p.count_instances) {
    n_inst = wave_sum_u32(n_inst); n_hq = wave_sum_u32(n_hq); n_foreign = wave_sum_u32(n_foreign);
    if (lane == 0) {
      if (n_inst) atomicAdd(reinterpret_cast<unsigned long long*>(&p.meta->instances), (unsigned long long)n_inst);
      if (n_hq) atomicAdd(reinterpret_cast<unsigned long long*>(&p.meta->hq_instances), (unsigned long long)n_hq);
      if (n_foreign) atomicAdd(reinterpret_cast<unsigned long long*>(&p.meta->foreign), (unsigned long long)n_foreign);
    }
  }
}

static void fill_common(qp_params& p, const qh_part_dst& d, qh_meta* meta, int check_owner, int count_instances) {
  p.out = d.out; p.out_cap = d.out_cap; p.out_counts = d.out_counts;
  p.out_count_stride = d.out_count_stride ? d.out_count_stride : 1;
  p.bin_shift = d.bin_shift; p.bin_mask = (1u << d.bin_bits) - 1; p.bin_bits = (uint32_t)d.bin_bits; p.out_bin_shift = d.out_bin_shift;
  p.spill = d.spill; p.spill_cap = d.spill_cap; p.meta = meta;
  p.check_owner = check_owner; p.count_instances = count_instances;
  p.out_pack_gsize = d.pack_gsize;
}

void qh_launch_partition_fastq(const qh_fastq_aux& a, uint32_t rec_cap, qh_meta* meta, const qh_geom& g,
                               const qh_part_dst& d, int check_owner, hipStream_t s) {
  qp_params p = {};
  fill_common(p, d, meta, check_owner, 1);
  p.out_pack_sbits = g.shard_bits;
  p.planes = a.planes; p.plane_words = a.plane_words; p.seg_desc = a.seg_desc;
  p.seg_total = a.tile_sums + (rec_cap + RB_TILE - 1) / RB_TILE;
  const dim3 grid(256), block(QP_THREADS);
  const int kclass = g.k < 4 ? 2 : g.k < QP_SEG ? 1 : 0;
#define QP_LAUNCH_FASTQ(OWNER, KC, KF) hipLaunchKernelGGL((k_partition<QP_SRC_FASTQ, OWNER, KC, KF>), grid, block, 0, s, p, g)
  static const char* nofix = getenv("QH_NO_KFIX");  // A/B tests
  static const char* nobinlo = getenv("QH_NO_BINLO");
  const int kfix = nofix ? 0 : g.k;
  const bool binlo = !nobinlo && d.bin_shift + d.bin_bits <= 31;
  if (!check_owner && kclass == 0 && binlo) {  // the unsharded build of ordinary mer lengths
    if (kfix == 24) hipLaunchKernelGGL((k_partition<QP_SRC_FASTQ, false, 0, 24, true>), grid, block, 0, s, p, g);
    else if (kfix == 31) hipLaunchKernelGGL((k_partition<QP_SRC_FASTQ, false, 0, 31, true>), grid, block, 0, s, p, g);
    else hipLaunchKernelGGL((k_partition<QP_SRC_FASTQ, false, 0, 0, true>), grid, block, 0, s, p, g);
  } else if (check_owner) {
    if (kclass == 2) QP_LAUNCH_FASTQ(true, 2, 0);
    else if (kclass == 1) QP_LAUNCH_FASTQ(true, 1, 0);
    else if (kfix == 24) QP_LAUNCH_FASTQ(true, 0, 24);
    else if (kfix == 31) QP_LAUNCH_FASTQ(true, 0, 31);
    else QP_LAUNCH_FASTQ(true, 0, 0);
  } else {
    if (kclass == 2) QP_LAUNCH_FASTQ(false, 2, 0);
    else if (kclass == 1) QP_LAUNCH_FASTQ(false, 1, 0);
    else if (kfix == 24) QP_LAUNCH_FASTQ(false, 0, 24);
    else if (kfix == 31) QP_LAUNCH_FASTQ(false, 0, 31);
    else QP_LAUNCH_FASTQ(false, 0, 0);
  }
#undef QP_LAUNCH_FASTQ
}

void qh_launch_partition_items(const uint64_t* in, uint64_t in_cap, const unsigned long long* in_counts,
                               uint32_t in_count_stride, uint32_t n_in_buckets, uint64_t in_n, int in_hashed, qh_meta* meta, const qh_geom& g,
                               const qh_part_dst& d, int check_owner, int count_instances, hipStream_t s) {
  qp_params p = {};
  fill_common(p, d, meta, check_owner, count_instances);
  p.out_pack_sbits = g.shard_bits;
  p.in = in; p.in_cap = in_cap; p.in_counts = in_counts; p.in_count_stride = in_count_stride ? in_count_stride : 1; p.in_n = in_n; p.in_hashed = in_hashed;
  const uint64_t most = in_counts ? in_cap : in_n;
  uint64_t tiles = (most + QP_TILE_ITEMS - 1) / QP_TILE_ITEMS;
  if (tiles < 1) tiles = 1;
  // one workgroup per CU is resident (144 KiB of LDS): grid-stride the tiles of a bucket
  uint64_t gx = n_in_buckets >= 256 ? 1 : (256 + n_in_buckets - 1) / n_in_buckets * 2;
  if (gx > tiles) gx = tiles;
  const dim3 grid((unsigned)gx, n_in_buckets), block(QP_THREADS);
  static const char* nobinlo = getenv("QH_NO_BINLO");
  const bool binlo = !nobinlo && d.bin_shift + d.bin_bits <= 31;
  const char* notall = getenv("QH_NO_TALL_TILE");  // A/B runs (read per launch)
  if (check_owner) hipLaunchKernelGGL((k_partition<QP_SRC_ITEMS, true, 0>), grid, block, 0, s, p, g);
  else if (binlo && d.bin_bits <= 10 && d.bin_bits >= 9 && !notall)
    hipLaunchKernelGGL((k_partition<QP_SRC_ITEMS, false, 0, 0, true, 18, 1024>), grid, block, 0, s, p, g);
  else if (binlo) hipLaunchKernelGGL((k_partition<QP_SRC_ITEMS, false, 0, 0, true>), grid, block, 0, s, p, g);
  else hipLaunchKernelGGL((k_partition<QP_SRC_ITEMS, false, 0>), grid, block, 0, s, p, g);
}

// ----------------------------------------------------------------- window build

// One workgroup per window. hash_with_quality::add (src/mer_database.hpp:94-113) on a window
// held in LDS, same slot format and probe rule as qh_insert.
//
// Every lane runs one insert chain: one LDS compare-and-swap per round, the CAS doubling as the
// read ("claim" rounds expect 0 and offer a fresh entry; "update" rounds expect the word the
// previous round returned). Probe lengths differ wildly between lanes, so a lane whose chain
// ends is handed the wave's next item at once: each wave owns a contiguous segment of the
// bucket, streams it 64 items at a time through a private LDS ring (queued when fewer than 64 are
// left in it, the next coalesced global load requested in the same step), and deals items to idle
// lanes by ballot rank. Nearly every CAS
// wave-instruction therefore carries 64 live lanes.
#define QW_RING 128  // items per wave ring (1 KiB, a power of two): at most 63 queued + 64 arriving

// NARROW: value and probe fields lie in the low 32 bits of the slot (vbits + pbits <= 32, true
// for every -b up to 24 at -p 126). Probing then only ever changes the low half of the word a
// lane offers, so the chain state is 32-bit and the round is ~25 VALU instructions. The kernel
// is VALU-issue bound (16-lane SIMDs, 4 waves each), so instructions per round are what count:
// ring bookkeeping is kept wave-uniform in SGPRs and the round is written without branches.
//
// A window is a TRANSACTION: when one of its items finds no slot within the reprobe limit, the window is not written
// back (the table keeps what it held before the pass, or zeros when its clearing was left to this pass), fail[w] is
// set and meta->win_failed counts it; the bucket stays as it was. The host then doubles the table and runs the failed
// windows' buckets again (FILTER: a window of the grown table takes, from the bucket of the window it descends from,
// the items whose hash names it). Nothing is queued per item, so a batch of any size needs no overflow memory.
// PACKED: the buckets hold 32-bit items {remainder | position | quality} (qh_part_dst::pack_gsize); the bucket number
// is the window the items were sorted for, which gives back the bits cut out: what the kernels see is (hash << 1 | q)
// as ever. pk_gsize is the gsize the buckets were made with (FILTER: the table has grown since).
template <bool PACKED>
__device__ __forceinline__ uint64_t qw_load(const uint64_t* __restrict__ items, uint64_t idx, uint32_t bkt, const qh_geom& g,
                                            int pk_gsize) {
  if (!PACKED) return items[idx];
  const uint32_t n = reinterpret_cast<const uint32_t*>(items)[idx];
  const uint64_t h = ((uint64_t)(n >> (QH_WINDOW_BITS + 1)) << pk_gsize) | ((uint64_t)bkt << (g.shard_bits + QH_WINDOW_BITS)) |
                     ((uint64_t)((n >> 1) & ((1u << QH_WINDOW_BITS) - 1u)) << g.shard_bits) | (uint32_t)g.shard_id;
  return (h << 1) | (n & 1u);
}

template <bool NARROW, bool FILTER, bool PACKED = false>
__global__ __launch_bounds__(QP_THREADS) void k_window_build(uint64_t* __restrict__ table, const qh_geom g,
                                                             const uint64_t* __restrict__ items, uint64_t cap,
                                                             const unsigned long long* __restrict__ counts,
                                                             int table_dirty, qh_meta* __restrict__ meta,
                                                             uint32_t* __restrict__ fail,
                                                             const uint32_t* __restrict__ prev_fail, uint32_t prev_mask,
                                                             uint32_t bucket_mask, uint32_t win_mask, int pk_gsize,
                                                             const uint32_t* __restrict__ wlist, uint32_t child_shift) {
  __shared__ uint64_t win[1u << QH_WINDOW_BITS];
  __shared__ uint64_t ring[QP_WAVES][QW_RING];  // window + rings = half of the CU's LDS: two workgroups per CU
  uint32_t* wnew = reinterpret_cast<uint32_t*>(&ring[0][0]);  // reused once the rings have drained
  // FILTER with a list: workgroup i runs child (i mod 2^child_shift) of the failed window wlist[i >> child_shift] of the
  // pass before (the windows w of the grown table with w & prev_mask == that window); every other window's flag was
  // cleared by the host
  uint32_t w = blockIdx.x;
  if (FILTER && wlist) {
    const uint32_t parent = wlist[blockIdx.x >> child_shift];
    if (parent == 0xFFFFFFFFu) return;  // (a list shorter than the host's count: nothing to run)
    w = parent + (blockIdx.x & ((1u << child_shift) - 1u)) * (prev_mask + 1u);
  }
  const uint32_t bkt = FILTER ? w & bucket_mask : w;
  if (FILTER && prev_fail[w & prev_mask] == 0) {  // the window this one descends from went through
    if (threadIdx.x == 0) fail[w] = 0;
    return;
  }
  uint64_t n = counts[bkt];
  if (n > cap) n = cap;
  if (n == 0) {  // nothing to add: the window stays as it is, or is cleared when the table's clearing was left to us
    if (table_dirty & 2) {
      uint4* z = reinterpret_cast<uint4*>(table + ((uint64_t)w << QH_WINDOW_BITS));
      for (uint32_t i = threadIdx.x; i < (1u << QH_WINDOW_BITS) / 2; i += QP_THREADS) z[i] = make_uint4(0, 0, 0, 0);
    }
    if (threadIdx.x == 0) fail[w] = 0;
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // this wave's segment of the bucket
  const uint32_t seg = (uint32_t)((n + QP_WAVES - 1) / QP_WAVES);
  const uint32_t s0 = (uint32_t)(((uint64_t)wave * seg) < n ? (uint64_t)wave * seg : n);
  const uint32_t seg_n = (uint32_t)((uint64_t)s0 + seg <= n ? seg : n - s0);
  const uint64_t sbase = (uint64_t)bkt * cap + s0;
  // first 64 items in flight while the window is brought into LDS; items [tail, tail + 64) are on request
  // whenever tail < seg_n
  uint64_t pend = (uint32_t)lane < seg_n ? qw_load<PACKED>(items, sbase + lane, bkt, g, pk_gsize) : QP_NONE;
  bool lane_fail = false;
  uint4* tw = reinterpret_cast<uint4*>(table + ((uint64_t)w << QH_WINDOW_BITS));
  uint4* lw = reinterpret_cast<uint4*>(win);
  const uint32_t nvec = (1u << QH_WINDOW_BITS) / 2;  // 16-byte pieces
  if (table_dirty & 1) {
#pragma unroll
    for (uint32_t i = 0; i < nvec / QP_THREADS; ++i) lw[i * QP_THREADS + tid] = tw[i * QP_THREADS + tid];
  } else {
#pragma unroll
    for (uint32_t i = 0; i < nvec / QP_THREADS; ++i) lw[i * QP_THREADS + tid] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();

  const int gsize = g.lsize + g.shard_bits;
  const uint32_t wmask = (1u << QH_WINDOW_BITS) - 1;
  const uint64_t vkeep = ~g.vmask, one_probe = 1ull << g.vbits;
  const uint32_t vkeep_lo = (uint32_t)vkeep, vmask_lo = (uint32_t)g.vmask, one_probe_lo = (uint32_t)one_probe;
  const uint32_t maxv_lo = (uint32_t)g.max_val;
  uint64_t* myring = ring[wave];
  // wave-uniform ring state: items dealt, items queued (positions modulo QW_RING), and the queue length below
  // which the next 64 are queued (0 once the segment has ended)
  uint32_t head = 0, tail = 0, gate = seg_n ? 64 : 0;
  uint32_t n_new = 0;
  // chain state
  bool live = false;
  uint64_t v = 0, want = 0, cmp = 0, nw = 0;                 // wide form
  uint32_t want_lo = 0, want_hi = 0, cmp_lo = 0, nw_lo = 0;  // narrow form
  uint32_t pos = 0, left = 0, q = 0;

  for (;;) {
    uint32_t avail = tail - head;
    if (avail < gate) {  // the ring runs low: queue the items that have arrived, request the next
      myring[(tail + lane) & (QW_RING - 1)] = pend;
      tail += seg_n - tail < 64 ? seg_n - tail : 64;
      if (tail < seg_n) pend = tail + lane < seg_n ? qw_load<PACKED>(items, sbase + tail + lane, bkt, g, pk_gsize) : QP_NONE;
      else gate = 0;
      avail = tail - head;  // >= 64 unless the segment has ended
    }
    __builtin_amdgcn_wave_barrier();
    // deal items to idle lanes
    const uint64_t need = __ballot(!live);
    if (avail == 0) {  // nothing queued means nothing left to queue
      if (need == ~0ull) break;
    } else {
      const uint32_t cnt = (uint32_t)__popcll((unsigned long long)need);
      const uint32_t give = cnt < avail ? cnt : avail;
      const uint32_t rank = lanes_below(need);
      if (!live && rank < give) {
        v = __hip_atomic_load(&myring[(head + rank) & (QW_RING - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint64_t h = v >> 1;
        // (FILTER: the bucket was made for a smaller table; items of this window's siblings stay where they are)
        const bool ours = !FILTER || ((uint32_t)((h >> g.shard_bits) >> QH_WINDOW_BITS) & win_mask) == w;
        pos = (uint32_t)(h >> g.shard_bits) & wmask;
        // want = (remainder | probe+1) << vbits: our key's slot word at this probe with a zero value
        want = (((gsize >= 64 ? 0 : (h >> gsize)) << g.pbits) | 1) << g.vbits;
        left = (uint32_t)g.max_reprobe;  // probes left after this one
        q = (uint32_t)v & 1u;            // a new entry is 2 | q (mer_database.hpp:105-110 from 0)
        if (NARROW) {
          want_lo = (uint32_t)want;
          want_hi = (uint32_t)(want >> 32);
          cmp_lo = 0;
          nw_lo = want_lo | 2u | q;
        } else {
          cmp = 0;
          nw = want | 2u | q;
        }
        live = ours;
      }
      head += give;
    }
    // one round: a single compare-and-swap per live lane
    if (live) {
      if (NARROW) {
        const uint32_t cmp_hi = cmp_lo ? want_hi : 0u;
        const uint64_t c = ((uint64_t)cmp_hi << 32) | cmp_lo;
        const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&win[pos]), (unsigned long long)c,
                                        (unsigned long long)(((uint64_t)want_hi << 32) | nw_lo));
        const uint32_t plo = (uint32_t)prev, phi = (uint32_t)(prev >> 32);
        const bool swapped = prev == c;
        const bool match = ((plo & vkeep_lo) == want_lo) & (phi == want_hi);  // our key lives here
        const uint32_t pv = plo & vmask_lo, phq = pv & 1u;
        // mer_database.hpp:102-111 on the value found
        const bool promote = phq < q;
        const bool can = promote | (((pv >> 1) != maxv_lo) & (phq == q));
        const uint32_t nv = promote ? 3u : pv + 2u;
        n_new += (uint32_t)(swapped & (cmp_lo == 0));
        const bool upd = !swapped & match & can;  // update round next
        const bool probe = !swapped & !match;     // another key: next probe
        lane_fail |= probe & (left == 0);         // reprobe limit: the window fails as a whole
        const bool go = probe & (left != 0);
        pos = go ? (pos + 1) & wmask : pos;
        left -= (uint32_t)go;
        want_lo += go ? one_probe_lo : 0u;
        cmp_lo = upd ? plo : 0u;
        nw_lo = want_lo | (upd ? nv : (2u | q));
        live = upd | go;
      } else {
        const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long*>(&win[pos]), (unsigned long long)cmp,
                                        (unsigned long long)nw);
        if (prev == cmp) {                          // the swap happened
          n_new += cmp == 0;
          live = false;
        } else if ((prev & vkeep) == want) {        // our key lives here: update round
          uint64_t nv;
          if (!qh_update_val(prev & g.vmask, q, g.max_val, &nv)) {
            live = false;                           // saturated / LQ instance of an HQ mer
          } else {
            cmp = prev;
            nw = want | nv;
          }
        } else if (left == 0) {                     // reprobe limit: the window fails as a whole
          lane_fail = true;
          live = false;
        } else {                                    // another key: next probe, claim round again
          --left;
          pos = (pos + 1) & wmask;
          want += one_probe;
          cmp = 0;
          nw = want | 2u | q;
        }
      }
    }
  }
  n_new = wave_sum_u32(n_new);
  const bool wave_fail = __ballot(lane_fail) != 0;
  __syncthreads();  // every chain has ended: the window is final and the rings are free
  if (lane == 0) { wnew[wave] = n_new; wnew[QP_WAVES + wave] = wave_fail ? 1u : 0u; }
  __syncthreads();
  bool failed = false;
#pragma unroll
  for (int i = 0; i < QP_WAVES; ++i) failed |= wnew[QP_WAVES + i] != 0;
  if (!failed) {
#pragma unroll
    for (uint32_t i = 0; i < nvec / QP_THREADS; ++i) tw[i * QP_THREADS + tid] = lw[i * QP_THREADS + tid];
  } else if (table_dirty & 2) {
#pragma unroll
    for (uint32_t i = 0; i < nvec / QP_THREADS; ++i) tw[i * QP_THREADS + tid] = make_uint4(0, 0, 0, 0);
  }
  if (tid == 0) {
    uint32_t t = 0;
    for (int i = 0; i < QP_WAVES; ++i) t += wnew[i];
    if (t && !failed) atomicAdd(reinterpret_cast<unsigned long long*>(&meta->new_keys), (unsigned long long)t);
    fail[w] = failed ? 1u : 0u;
    if (failed) atomicAdd(&meta->win_failed, 1u);
  }
}

// ---- counting form of the window build
//
// hash_with_quality::add is order independent: a key's final value is (min(H, max), 1) when it
// was seen H > 0 times with quality, else (min(L, max), 0) for its L low-quality sightings
// (src/mer_database.hpp:102-111 applied in any order). While a window sits in LDS its slots are
// therefore kept as [tag | H | L] with two plain counters of `cb` bits, tag = remainder | probe+1
// as in the table, and a sighting is ONE returning 64-bit add at the probe position:
//   - the old word carries our tag: done (three sightings out of four at 30x coverage);
//   - it carries no tag: the slot is free, claim it by compare-and-swap of the whole word
//     (retried while other lanes' counters move; if the same key wins the race our count is
//     already in, if another key wins we withdraw);
//   - it carries another tag: withdraw the increment with a second add and probe on.
// Withdrawn increments are only ever visible in the counter fields, which nobody reads before the
// final barrier, and counters cannot reach the tag (2^cb > bucket capacity + max count, checked
// by the launcher). The words are converted from / to the table's [tag | count << 1 | hq] form
// when the window is loaded / stored. Used when tag bits + 2 * cb fit 64 with 2^cb above the bucket
// capacity (k = 31 reaches that at the table sizes of BASELINE configs[3]: 35 tag bits, cb = 14); other
// geometries take k_window_build.
//
// CB16: cb == 16, i.e. the tag is the high word and both counters lie in the low word (no counter carry ever
// crosses bit 32: H cannot overflow). Tag tests, claims and the chain state are then 32-bit operations. The
// kernel is bound by the instructions a wave issues per round (one workgroup of 16 waves per CU, LDS, VALU
// and HBM each about half used), so the round is kept short: a 128-entry ring whose refill and next request
// are one step, liveness folded into the probe counter, the increment decoded once per item.
//
// GUARD: counters narrower than a bucket (k = 31 at tables below 2^32 slots: 41 tag bits leave 11-bit counters for
// buckets of thousands). A counter moves by one per add or withdrawal, so before a field can carry into its neighbour
// some add must return `hot_limit` = 2^cb - 1 - 1024 (every lane of the workgroup has at most one add in flight): the
// lane that sees it marks the window as failed-hot (fail[w] = 2) and the window is left as it was, like one that ran
// out of probes; the host runs such windows again with the compare-and-swap kernel, at the same table size. A key
// must be seen a thousand times in one bucket for that: the reads' own coverage never does it, repeats may.
template <bool CB16, bool GUARD = false, bool PACKED = false>
__global__ __launch_bounds__(QP_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_window_count(uint64_t* __restrict__ table, const qh_geom g,
                                                             const uint64_t* __restrict__ items, uint64_t cap,
                                                             const unsigned long long* __restrict__ counts,
                                                             uint32_t n_windows, int table_dirty,
                                                             qh_meta* __restrict__ meta, int cb_arg,
                                                             uint32_t* __restrict__ fail, uint32_t hot_limit, int pk_gsize) {
  typedef typename std::conditional<CB16, uint32_t, uint64_t>::type tag_t;  // tag = remainder | probe + 1
  __shared__ uint64_t win[1u << QH_WINDOW_BITS];
  __shared__ uint64_t ring[QP_WAVES][QW_RING];  // window + rings = half of the CU's LDS: two workgroups per CU
  uint32_t* wnew = reinterpret_cast<uint32_t*>(&ring[0][0]);  // reused after the last window
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int cb = CB16 ? 16 : cb_arg;
  const int tagshift = 2 * cb;
  const uint64_t cmask = (1ull << cb) - 1;
  uint4* lw = reinterpret_cast<uint4*>(win);
  const uint32_t nvec = (1u << QH_WINDOW_BITS) / 2;  // 16-byte pieces
  auto to_counting = [&](uint64_t slot) -> uint64_t {
    const uint64_t v = slot & g.vmask, c = v >> 1;
    return ((slot >> g.vbits) << tagshift) | ((v & 1) ? c << cb : c);
  };
  auto to_table = [&](uint64_t word) -> uint64_t {
    const uint64_t t = word >> tagshift, hc = (word >> cb) & cmask, lc = word & cmask;
    const uint64_t c = hc ? hc : lc;
    const uint64_t v = ((c < g.max_val ? c : g.max_val) << 1) | (hc ? 1u : 0u);
    return t ? (t << g.vbits) | v : 0ull;
  };
  const int gsize = g.lsize + g.shard_bits;
  const uint32_t wmask = (1u << QH_WINDOW_BITS) - 1;
  uint64_t* myring = ring[wave];
  uint32_t n_new = 0;

  // The workgroup walks windows blockIdx.x, + gridDim.x, ... (one workgroup per CU: the window fills its LDS), so
  // that what a fresh workgroup would wait for is requested a window ahead: the next bucket's size while this
  // window is counted, the first items of this wave's segment of it before this window is written out.
  uint32_t seg_n = 0;             // this wave's segment of the bucket: items [sbase, sbase + seg_n) of bucket sbkt
  uint64_t sbase = 0;
  // items [tail, tail + 64) of the segment, requested whenever tail < seg_n. PACKED: the 32-bit word as it lies in the
  // bucket -- this kernel needs no more of an item than the word holds (the buckets were made for this table: pk_gsize is
  // its gsize), so nothing is put together again
  typedef typename std::conditional<PACKED, uint32_t, uint64_t>::type item_t;
  const item_t* __restrict__ ibase = reinterpret_cast<const item_t*>(items);
  item_t pend = (item_t)QP_NONE;
  auto open_bucket = [&](uint32_t w, uint64_t n) {
    const uint32_t seg = (uint32_t)((n + QP_WAVES - 1) / QP_WAVES);
    const uint32_t s0 = (uint32_t)(((uint64_t)wave * seg) < n ? (uint64_t)wave * seg : n);
    seg_n = (uint32_t)((uint64_t)s0 + seg <= n ? seg : n - s0);
    sbase = (uint64_t)w * cap + s0;
    pend = (uint32_t)lane < seg_n ? ibase[sbase + lane] : (item_t)QP_NONE;
  };
  auto bucket_size = [&](uint32_t w) -> uint64_t {
    if (w >= n_windows) return 0;
    const uint64_t n = counts[w];
    return n > cap ? cap : n;
  };
  uint32_t w = blockIdx.x;
  uint64_t n = bucket_size(w);
  open_bucket(w, n);

  for (; w < n_windows; w += gridDim.x) {
    const uint64_t n_next = bucket_size(w + gridDim.x);
    uint4* tw = reinterpret_cast<uint4*>(table + ((uint64_t)w << QH_WINDOW_BITS));
    if (n == 0) {  // nothing to add: the window stays as it is, or is cleared when the table's clearing was left to us
      if (table_dirty & 2)
        for (uint32_t i = tid; i < nvec; i += QP_THREADS) tw[i] = make_uint4(0, 0, 0, 0);
      if (tid == 0) fail[w] = 0;
      n = n_next;
      open_bucket(w + gridDim.x, n);
      continue;
    }
    // The window's flag starts at 0 here; a lane whose item runs out of probes sets it (a plain store, on a path
    // nearly never taken) and every wave reads it back after the window's last barrier: the failure costs the
    // loop no register and no instruction.
    if (tid == 0) __hip_atomic_store(&fail[w], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // every thread fills the pieces of the window it wrote out for the previous one: no barrier in between
    if (table_dirty & 1) {
#pragma unroll
      for (uint32_t i = 0; i < nvec / QP_THREADS; ++i) {
        const uint4 x = tw[i * QP_THREADS + tid];
        const uint64_t a = to_counting(((uint64_t)x.y << 32) | x.x), b = to_counting(((uint64_t)x.w << 32) | x.z);
        lw[i * QP_THREADS + tid] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
      }
    } else {
#pragma unroll
      for (uint32_t i = 0; i < nvec / QP_THREADS; ++i) lw[i * QP_THREADS + tid] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    // wave-uniform ring state: items dealt, items queued (positions modulo QW_RING), and the queue length below
    // which the next 64 are queued (0 once the segment has ended)
    uint32_t head = 0, tail = 0, gate = seg_n ? 64 : 0;
    // chain state: tag at this probe, position, increment (1, or 1 << cb with quality), probes left (< 0: no chain)
    tag_t T = 0;
    uint32_t pos = 0, inc = 1;
    int left = -1;
    const uint32_t n_new_before = n_new;  // (GUARD: a failed window's LDS image may be corrupt: its claims are forgotten)

    for (;;) {
      uint32_t avail = tail - head;
      if (avail < gate) {
        // decoded once, with every lane busy. 16-bit counters: {tag at probe 0 : 32 | log2(increment) << 16 | position};
        // wide tags: {tag at probe 0 : up to 50 | quality : 1 | position : QH_WINDOW_BITS}
        uint32_t p0;
        uint64_t t0;
        if (PACKED) {
          p0 = ((uint32_t)pend >> 1) & wmask;
          t0 = ((uint64_t)((uint32_t)pend >> (QH_WINDOW_BITS + 1)) << g.pbits) | 1u;
        } else {
          const uint64_t h = pend >> 1;
          p0 = (uint32_t)(h >> g.shard_bits) & wmask;
          t0 = ((gsize >= 64 ? 0 : (h >> gsize)) << g.pbits) | 1u;
        }
        const uint32_t sh = (uint32_t)pend & 1u ? (uint32_t)cb : 0u;
        myring[(tail + lane) & (QW_RING - 1)] =
            CB16 ? ((t0 << 32) | p0 | (sh << 16)) : ((t0 << (QH_WINDOW_BITS + 1)) | ((uint64_t)((uint32_t)pend & 1u) << QH_WINDOW_BITS) | p0);
        tail += seg_n - tail < 64 ? seg_n - tail : 64;
        if (tail < seg_n) pend = tail + lane < seg_n ? ibase[sbase + tail + lane] : (item_t)QP_NONE;
        else gate = 0;
        avail = tail - head;  // >= 64 unless the segment has ended
      }
      __builtin_amdgcn_wave_barrier();
      const uint64_t need = __ballot(left < 0);
      if (avail == 0) {  // nothing queued means nothing left to queue
        if (need == ~0ull) break;
      } else {
        const uint32_t cnt = (uint32_t)__popcll((unsigned long long)need);
        const uint32_t give = cnt < avail ? cnt : avail;
        const uint32_t rank = lanes_below(need);
        if (left < 0 && rank < give) {
          const uint64_t e = __hip_atomic_load(&myring[(head + rank) & (QW_RING - 1)], __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
          if (CB16) {
            T = (tag_t)(e >> 32);
            inc = 1u << (((uint32_t)e >> 16) & 31u);
          } else {
            T = (tag_t)(e >> (QH_WINDOW_BITS + 1));
            inc = ((uint32_t)e >> QH_WINDOW_BITS) & 1u ? 1u << cb : 1u;
          }
          pos = (uint32_t)e & wmask;
          left = (int)g.max_reprobe;
        }
        head += give;
      }
      if (left >= 0) {
        unsigned long long* slot = reinterpret_cast<unsigned long long*>(&win[pos]);
        const uint64_t old = atomicAdd(slot, (unsigned long long)inc);
        if (GUARD) {
          const uint32_t field = (inc == 1u ? (uint32_t)old : (uint32_t)(old >> cb)) & (uint32_t)cmask;
          if (__builtin_expect(field >= hot_limit, 0))
            __hip_atomic_store(&fail[w], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        tag_t t = CB16 ? (tag_t)(old >> 32) : (tag_t)(old >> tagshift);
        if (t == 0) {  // a free slot: claim it, sighting included
          // Tried again in place while only counters move: every failure is then another chain's add or
          // withdrawal, of which there are finitely many. (Withdrawing and coming back in the next round
          // instead lets two waves in step fail each other's compare-and-swap for seconds.)
          uint64_t expect = old + inc;
          auto mine = [&](uint64_t e) -> uint64_t {
            return CB16 ? (((uint64_t)T << 32) | (uint32_t)e) : (e | ((uint64_t)T << tagshift));
          };
          uint64_t prev = atomicCAS(slot, (unsigned long long)expect, (unsigned long long)mine(expect));
          bool claimed = prev == expect;
          t = CB16 ? (tag_t)(prev >> 32) : (tag_t)(prev >> tagshift);
          if (__builtin_expect(!claimed && t == 0, 0)) {
            do {
              expect = prev;
              prev = atomicCAS(slot, (unsigned long long)expect, (unsigned long long)mine(expect));
              claimed = prev == expect;
              t = CB16 ? (tag_t)(prev >> 32) : (tag_t)(prev >> tagshift);
            } while (!claimed && t == 0);
          }
          if (claimed) {
            ++n_new;
            t = T;
          }
        }
        if (t == T) {
          left = -1;  // counted
        } else {
          // another key lives here: withdraw and probe on
          __hip_atomic_fetch_sub(slot, (unsigned long long)inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (__builtin_expect(left == 0, 0))  // reprobe limit: the window fails as a whole (see k_window_build)
            __hip_atomic_store(&fail[w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          --left;
          pos = (pos + 1) & wmask;
          ++T;
        }
      }
    }
    __syncthreads();  // every chain of the window has ended (and every store to its flag has been made)
    const uint32_t fail_kind = __hip_atomic_load(&fail[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool failed = fail_kind != 0;
    n = n_next;
    open_bucket(w + gridDim.x, n);  // the next bucket's first items travel while this window is written out
    typedef uint32_t qh_v4u __attribute__((ext_vector_type(4)));
    if (__builtin_expect(!failed, 1)) {
#pragma unroll
      for (uint32_t i = 0; i < nvec / QP_THREADS; ++i) {
        const uint4 x = lw[i * QP_THREADS + tid];
        const uint64_t a = to_table(((uint64_t)x.y << 32) | x.x), b = to_table(((uint64_t)x.w << 32) | x.z);
        const qh_v4u o = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
        __builtin_nontemporal_store(o, reinterpret_cast<qh_v4u*>(tw + i * QP_THREADS + tid));
      }
    } else {
      // the window stays as it was. The slots claimed in it were counted as new keys: taken back here (tags in
      // LDS now, minus the entries the window held when it was loaded), the pass that runs the bucket again
      // counts them when they stay
      if (GUARD) {
        n_new = n_new_before;
      } else {
        int32_t claimed = 0;
        for (uint32_t i = tid; i < (1u << QH_WINDOW_BITS); i += QP_THREADS) {
          claimed += (win[i] >> tagshift) != 0 ? 1 : 0;
          if (table_dirty & 1) claimed -= table[((uint64_t)w << QH_WINDOW_BITS) + i] != 0 ? 1 : 0;
        }
        atomicAdd(reinterpret_cast<unsigned long long*>(&meta->new_keys), (unsigned long long)(long long)(-claimed));
      }
      if (table_dirty & 2)
        for (uint32_t i = tid; i < nvec; i += QP_THREADS) tw[i] = make_uint4(0, 0, 0, 0);
      if (tid == 0) atomicAdd(fail_kind == 2u ? &meta->win_hot : &meta->win_failed, 1u);
      __syncthreads();  // the window in LDS has been read by all before the next one is loaded over it
    }
  }
  n_new = wave_sum_u32(n_new);
  __syncthreads();  // the rings are free
  if (lane == 0) wnew[wave] = n_new;
  __syncthreads();
  if (tid == 0) {
    uint32_t t = 0;
    for (int i = 0; i < QP_WAVES; ++i) t += wnew[i];
    if (t) atomicAdd(reinterpret_cast<unsigned long long*>(&meta->new_keys), (unsigned long long)t);
  }
}

uint64_t qh_window_count_cap_limit(const qh_geom& g) {
  const int tagbits = g.rbits + g.pbits;
  int cb = (64 - tagbits) / 2;
  if (cb > 31) cb = 31;
  if (tagbits > 64 - (QH_WINDOW_BITS + 1) || cb < 10 || (1ull << cb) <= g.max_val + 2) return 0;  // compare-and-swap kernel
  if (tagbits <= 32 && 65536ull > g.max_val + 2 + 1024) return 65536ull - g.max_val - 2;          // 16-bit counters
  if (cb >= 11 && (1ull << cb) > 1025 + 2 * (g.max_val + 1)) return 0;                             // watched counters: any bucket
  return (1ull << cb) - g.max_val - 2;
}

void qh_launch_window_build(uint64_t* table, const qh_geom& g, const uint64_t* items, uint64_t cap,
                            const unsigned long long* counts, uint32_t n_windows, int table_dirty, uint32_t* fail,
                            qh_meta* meta, hipStream_t s, int pack_gsize) {
  // counting form when [tag | H | L] fits 64 bits with counters that can hold a whole bucket
  const char* force = getenv("QH_WINDOW_KERNEL");  // A/B tests: "cas" compare-and-swap kernels, "count" generic counters
  const int tagbits = g.rbits + g.pbits;
  int cb = (64 - tagbits) / 2;
  if (cb > 31) cb = 31;
  // (the tag must also fit a ring entry beside the position and the quality bit; 2^cb > max_val is implied)
  const bool ring_ok = tagbits <= 64 - (QH_WINDOW_BITS + 1);
  const bool fits = ring_ok && cb >= 10 && (1ull << cb) > cap + g.max_val + 1;
  // counters narrower than a bucket, but wide enough to be watched (see GUARD)
  const bool guard = !fits && ring_ok && cb >= 11 && (1ull << cb) > 1025 + 2 * (g.max_val + 1) && !(force && !strcmp(force, "noguard"));
  const bool pk = pack_gsize > 0;
  if ((fits || guard) && !(force && !strcmp(force, "cas"))) {
    // one workgroup per CU walks the windows (see the kernel): QH_WINDOW_GRID overrides the grid for A/B tests
    const char* gs = getenv("QH_WINDOW_GRID");
    uint32_t grid = gs ? (uint32_t)atoi(gs) : 512u;
    if (grid == 0 || grid > n_windows) grid = n_windows;
    // 16-bit counters: the tag is the high word ("count": the generic form even then)
    const bool cb16 = fits && tagbits <= 32 && 65536ull > cap + g.max_val + 1 && !(force && !strcmp(force, "count"));
#define QW_COUNT(CB16, GUARD, PACKED, CB, HOT)                                                                            \
  hipLaunchKernelGGL((k_window_count<CB16, GUARD, PACKED>), dim3(grid), dim3(QP_THREADS), 0, s, table, g, items, cap, counts, \
                     n_windows, table_dirty, meta, CB, fail, HOT, pack_gsize)
    const uint32_t hot = (uint32_t)((1u << cb) - 1u - 1024u);
    if (cb16) { if (pk) QW_COUNT(true, false, true, 16, 0u); else QW_COUNT(true, false, false, 16, 0u); }
    else if (fits) { if (pk) QW_COUNT(false, false, true, cb, 0u); else QW_COUNT(false, false, false, cb, 0u); }
    else { if (pk) QW_COUNT(false, true, true, cb, hot); else QW_COUNT(false, true, false, cb, hot); }
#undef QW_COUNT
    return;
  }
#define QW_BUILD(NARROW, PACKED)                                                                                          \
  hipLaunchKernelGGL((k_window_build<NARROW, false, PACKED>), dim3(n_windows), dim3(QP_THREADS), 0, s, table, g, items, cap, \
                     counts, table_dirty, meta, fail, nullptr, 0u, 0u, 0u, pack_gsize, nullptr, 0u)
  if (g.vbits + g.pbits <= 32) { if (pk) QW_BUILD(true, true); else QW_BUILD(true, false); }
  else { if (pk) QW_BUILD(false, true); else QW_BUILD(false, false); }
#undef QW_BUILD
}

// the windows whose flag is set, in any order; list[cap] counts them (list preset to ~0 by the caller)
__global__ __launch_bounds__(256) void k_failed_windows(const uint32_t* __restrict__ fail, uint32_t n,
                                                        uint32_t* __restrict__ list, uint32_t cap) {
  for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < n; w += gridDim.x * 256) {
    if (fail[w]) {
      const uint32_t at = atomicAdd(&list[cap], 1u);
      if (at < cap) list[at] = w;
    }
  }
}

// The failed windows of a pass again, after the table has grown (or, for windows that only met a hot key, at the same
// size): buckets made for n_buckets windows, table of n_windows (a power-of-two multiple); prev_fail has the flags of
// the pass before (n_prev windows), n_failed of them set. `list` (n_failed + 1 words) receives the failed windows, and
// only their descendants get a workgroup: a launch over all 2^16 windows of a 2^29-slot table took 0.18 ms to find
// the handful that had anything to do.
void qh_launch_window_redo(uint64_t* table, const qh_geom& g, const uint64_t* items, uint64_t cap,
                           const unsigned long long* counts, uint32_t n_buckets, uint32_t n_windows, uint32_t* fail,
                           const uint32_t* prev_fail, uint32_t n_prev, uint32_t n_failed, uint32_t* list, qh_meta* meta,
                           hipStream_t s, int pack_gsize) {
  hipError_t e = hipMemsetAsync(list, 0xFF, (size_t)n_failed * 4, s);
  if (e == hipSuccess) e = hipMemsetAsync(list + n_failed, 0, 4, s);
  if (e == hipSuccess) e = hipMemsetAsync(fail, 0, (size_t)n_windows * 4, s);
  (void)e;  // (the caller checks hipGetLastError)
  uint32_t cg = (n_prev + 255) / 256;
  if (cg > 1024) cg = 1024;
  hipLaunchKernelGGL(k_failed_windows, dim3(cg), dim3(256), 0, s, prev_fail, n_prev, list, n_failed);
  uint32_t child_shift = 0;
  while ((n_prev << child_shift) < n_windows) ++child_shift;
  const dim3 grid(n_failed << child_shift);
  if (pack_gsize > 0)
    hipLaunchKernelGGL((k_window_build<false, true, true>), grid, dim3(QP_THREADS), 0, s, table, g, items, cap, counts,
                       1, meta, fail, prev_fail, n_prev - 1u, n_buckets - 1u, n_windows - 1u, pack_gsize, list, child_shift);
  else
    hipLaunchKernelGGL((k_window_build<false, true, false>), grid, dim3(QP_THREADS), 0, s, table, g, items, cap, counts,
                       1, meta, fail, prev_fail, n_prev - 1u, n_buckets - 1u, n_windows - 1u, 0, list, child_shift);
}
