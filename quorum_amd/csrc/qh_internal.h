// qh_internal.h -- handle definitions shared by qh_build.hip and qh_query.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>
#include "qh_host.h"
#include "qh_kernels.h"

struct qh_build {
  int device = 0;
  hipStream_t stream = nullptr;      // stream all work is enqueued on
  hipStream_t own_stream = nullptr;  // the handle's own stream
  qh_geom g;
  int reprobe_limit = 126;
  uint64_t* table = nullptr;
  uint8_t* d_chunk = nullptr;  size_t chunk_cap = 0;  // staging of host-fed chunks
  uint4* d_recs = nullptr;     size_t rec_cap = 0;    // record table of the chunk
  uint32_t* d_tiles = nullptr; size_t tile_cap = 0;   // newline counts / prefix per tile
  ushort4* d_nlplane = nullptr; size_t nlplane_cap = 0;  // newline bits per 16 chunk bytes (+ what follows them)
  uint64_t* d_ovf[2] = {nullptr, nullptr}; size_t ovf_cap[2] = {0, 0};  // overflow lists (bytes)
  uint64_t* d_items = nullptr; size_t items_cap = 0;  // staging of host-fed items
  uint64_t* d_shard_counts = nullptr;
  // radix-partitioned build: block prefix of the record table, bucket levels
  uint32_t* d_rbsums = nullptr;    size_t rbsums_cap = 0;
  uint2* d_segdesc = nullptr;      size_t segdesc_cap = 0;  // per-segment descriptors
  uint64_t* d_planes = nullptr;    size_t planes_cap = 0;   // bit planes of the chunk
  uint64_t* d_bins[2] = {nullptr, nullptr}; size_t bins_cap[2] = {0, 0};       // level-1 / window buckets (bytes)
  unsigned long long* d_bin_counts[2] = {nullptr, nullptr}; size_t bin_counts_cap[2] = {0, 0};
  // The other set of the per-chunk FASTQ arrays above (record table, tile counts, newline lists, segment sums and
  // descriptors, text planes): while a partitioned chunk's split kernels still run, the NEXT chunk's index and segment
  // kernels fill this set on a stream of their own (qh_build_add_fastq_device), then the two sets change places
  struct prep_set {
    uint4* recs = nullptr; size_t rec_cap = 0;
    uint32_t* tiles = nullptr; size_t tile_cap = 0;
    ushort4* nlplane = nullptr; size_t nlplane_cap = 0;
    uint32_t* rbsums = nullptr; size_t rbsums_cap = 0;
    uint2* segdesc = nullptr; size_t segdesc_cap = 0;
    uint64_t* planes = nullptr; size_t planes_cap = 0;
  } alt;
  hipStream_t pstream = nullptr;     // stream of the early preparation
  bool reset_pending = false;                // qh_build_reset was called with a chunk in flight: applied when that chunk is collected
  hipEvent_t prep_done = nullptr;
  hipEvent_t main_mark = nullptr;            // on the main stream: the point of the chunk in flight after which the next chunk's early preparation may start
  qh_meta* d_pmeta = nullptr;        // its own counters (lines, records, format verdict), copied into d_meta afterwards
  bool last_partitioned = false;     // the chunk in flight went through the partitioned build
  bool zero_pending = false;  // qh_build_reset left the clearing of the table to the next partitioned chunk
  bool table_dirty = false;   // the table holds entries (a window must be loaded before it is updated)
  int mode = 0;               // 0 auto, 1 direct global-CAS build, 2 radix-partitioned LDS build
  int level1_bits = -1;       // -1 auto: buckets of the first partition level (QH_LEVEL1_BITS)
  int cur_ovf = 0;
  // Deferred window pass: the window buckets (d_bins[1]) of successive partitioned chunks accumulate in HBM and the
  // table is streamed once per BATCH of chunks instead of once per chunk (qh_build.hip: run_partitioned / launch_flush)
  long long defer_budget = -1;   // bytes the window buckets of a batch may take: -1 automatic, 0 = every chunk is flushed
  size_t free_at_first_use = 0;  // free device memory when the first batch was sized (automatic budget)
  bool defer_open = false;       // the window buckets hold items that are not in the table yet
  size_t defer_capw = 0;         // capacity of a window bucket in the open batch
  int defer_pack = 0;                        // > 0: the open batch's buckets hold packed 32-bit items made with this gsize (qh_part_dst::pack_gsize)
  uint32_t defer_nw = 0;         // windows of the table the batch was opened for
  uint64_t defer_items = 0;      // items of the batch's chunks collected so far
  uint64_t defer_cap_items = 0;  // items the batch takes before it is flushed
  uint64_t* defer_spill = nullptr; uint64_t defer_spill_cap = 0;  // spill list of the batch's last chunk
  uint64_t* d_bspill = nullptr; size_t bspill_cap = 0;           // items that overflowed a bucket, of the whole batch (bytes)
  uint64_t bspill_n = 0;
  uint32_t* d_wfail[2] = {nullptr, nullptr}; size_t wfail_cap[2] = {0, 0};  // per window: the pass left it as it was
  uint32_t* d_wlist = nullptr; size_t wlist_cap = 0;  // the failed windows of a pass, for the pass that runs them again
  bool pending_deferred = false; // the chunk in flight went into the open batch
  size_t pending_fastq_len = 0;  // bytes of the FASTQ chunk in flight (items per byte are learnt at collection)
  double items_per_byte = 0;     // largest ratio seen so far: sizes a batch better than the len / 2 bound
  qh_meta* d_meta = nullptr;
  qh_meta* h_meta = nullptr;  // pinned mirror
  qh_build_stats st;
  bool pending = false;       // kernels of a chunk are in flight, meta not yet collected
  // optional timing of the dominant kernel (insert/build) with HIP events on `stream`
  bool timing = false;
  hipEvent_t evk[4][2] = {};                 // per timed kernel: begin/end
  bool evk_pending[4] = {false, false, false, false};
  hipEvent_t evp[2][2] = {};                 // the segment kernels of early preparations (second stream), two pairs in turn: counted as kernel 0
  bool evp_pending[2] = {false, false};
  int evp_cur = 0;
  double kernel_ms[4] = {0, 0, 0, 0};        // 0 direct/insert, 1 partition L1 (+extraction), 2 partition L2, 3 window build
  uint64_t kernel_launches[4] = {0, 0, 0, 0};
  void* noalloc_scratch = nullptr;  // this device's copy of qh_build.hip's file-scope scratch array
  int failed = 0;             // sticky fatal status (QH_EFULL ...)
  std::string failed_msg;
};

struct qh_query {
  int device = 0;
  qh_geom g;
  uint64_t* table = nullptr;
  bool owns_table = true;
  hipStream_t stream = nullptr;
  qh_meta* d_meta = nullptr;
  qh_meta* h_meta = nullptr;
  // staging for the host-pointer entry points of the table reductions (stats, histogram, dump): one caller at a time
  void* d_out = nullptr; size_t out_cap = 0;
  std::mutex mu;
  // The batched lookups (operator[], get_val, get_best_alternatives, is_contaminant) are const and called concurrently by
  // the corrector's threads upstream (src/mer_database.hpp:284, src/error_correct_reads.cc:114,173-175): every call takes
  // a LANE of its own -- a stream and staging buffers -- so that concurrent callers overlap on the GPU instead of
  // queueing on one mutex. Lanes are created on demand, QH_QUERY_MAX_LANES at most; further callers wait for a free one.
  struct lane {
    hipStream_t stream = nullptr;
    void* d_in = nullptr;  size_t in_cap = 0;
    void* d_in2 = nullptr; size_t in2_cap = 0;
    void* d_out = nullptr; size_t out_cap = 0;
    bool busy = false;
  };
  std::vector<lane*> lanes;
  std::mutex lane_mu;
  std::condition_variable lane_cv;
};
#define QH_QUERY_MAX_LANES 16

int qh_query_from_table(int device, uint64_t* table, const qh_geom& g, qh_query** out);
int qh_table_dump(int device, hipStream_t stream, const uint64_t* table, const qh_geom& g, qh_meta* d_meta,
                  qh_meta* h_meta, uint64_t* mers, uint64_t* vals, uint64_t cap, uint64_t* n);
int qh_ensure_device(void** p, size_t* cap, size_t need, const char* what);

// Sequential FASTQ parser with jellyfish::whole_sequence_parser's rules (qh_build.hip). hdrs (optional)
// receives the header text of every record: offset of the byte after '@' and its length.
struct qh_hdr { uint32_t off, len; };
int qh_parse_fastq_host(const char* buf, size_t len, bool final_chunk, std::vector<qh_record>& recs,
                        std::string& scratch, size_t* consumed, std::vector<qh_hdr>* hdrs);

extern "C" int qh_build_write_shard(qh_build* h, const char* path, int argc, const char* const* argv, int write_header);
