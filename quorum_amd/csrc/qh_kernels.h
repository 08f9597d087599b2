// qh_kernels.h -- launchers of the gfx950 kernels (qh_kernels.hip). Internal to
// libquorum_hip; the public surface is include/quorum_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "qh_spec.h"

// Device-resident per-chunk state; mirrored to pinned host memory by qh_build_sync.
struct qh_meta {
  uint32_t n_lines;      // lines found in the chunk (incl. a virtual last newline)
  uint32_t nrec;         // records of the chunk
  uint32_t format_err;   // != 0: chunk is not strict 4-line FASTQ; nothing inserted
  uint32_t first_bad;    // smallest offending record / line index
  uint32_t ovf_lost;     // != 0: overflow list itself overflowed (fatal)
  uint32_t rehash_fail;  // != 0: a key did not fit the grown table
  uint64_t ovf_count;    // items that hit the reprobe limit (waiting for growth)
  uint64_t spill_count;  // items that did not fit their bucket / window (inserted directly afterwards)
  uint64_t instances;    // counted_low
  uint64_t hq_instances; // counted_high
  uint64_t new_keys;     // slots claimed
  uint64_t foreign;      // items owned by another shard (dropped)
  uint64_t dump_count;   // cursor of k_dump
  uint64_t stat_distinct, stat_total;
  uint32_t win_failed;   // windows of the last pass that were left as they were (an item hit the reprobe limit)
  uint32_t win_hot;      // ... because a counter came near its width (k_window_count's GUARD): same table size again
  uint64_t stat_occupied;  // k_stats: slots that hold an entry of either quality
};

enum { QH_FMT_LINES_MOD4 = 1, QH_FMT_NO_AT = 2, QH_FMT_NO_PLUS = 4, QH_FMT_LEN_MISMATCH = 8,
       QH_FMT_TOO_MANY = 16, QH_FMT_EMPTY_TAIL = 32, QH_FMT_LINE_TOO_LONG = 64 };

#define QH_TILE_BYTES 16384u

// --- FASTQ chunk -> record table -------------------------------------------------
// rec_cap = capacity of `recs` in records. tile_counts has ceil(len/QH_TILE_BYTES)+1 entries,
// nl_plane: room for the tiles' newline lists, 4096 16-bit entries a tile = (len/16 + 2) * 8 + 8192 bytes. planes != null: also fill the text planes of qh_fastq_aux
// (same pass over the text).
void qh_launch_index_fastq(const uint8_t* buf, size_t len, uint32_t* tile_counts, ushort4* nl_plane, uint4* recs,
                           uint32_t rec_cap, qh_meta* meta, const struct qh_fastq_aux* planes, int min_qual,
                           hipStream_t s);

// --- record table -> k-mers -> table (direct global-CAS build) --------------------
void qh_launch_build_direct(const uint8_t* buf, const uint4* recs, qh_meta* meta, const qh_geom& g,
                            uint64_t* table, uint64_t* ovf, uint64_t ovf_cap, hipStream_t s);

// packed items -> table. hashed = 0: (mer<<1|q) as the C ABI passes them; 1: the
// internal (qh_mix(mer)<<1|q) form of the overflow / spill lists. n_dev (optional) =
// device-side count capped by n. count_instances = 0 when re-inserting items that
// were already counted when first seen.
void qh_launch_insert_items(const uint64_t* items, uint64_t n, const unsigned long long* n_dev,
                            int count_instances, int hashed, qh_meta* meta, const qh_geom& g, uint64_t* table,
                            uint64_t* ovf, uint64_t ovf_cap, hipStream_t s);

// old table -> new (empty) table, carrying the value fields (mer_database.hpp:165-176)
void qh_launch_rehash(const uint64_t* old_table, const qh_geom& og, uint64_t* new_table,
                      const qh_geom& ng, qh_meta* meta, hipStream_t s);

// n distinct (mer, value) pairs -> table (no shards); meta->rehash_fail when a key found no place
void qh_launch_insert_valued(const uint64_t* mers, const uint64_t* vals, uint64_t n, uint64_t* table, const qh_geom& g,
                             qh_meta* meta, hipStream_t s);

// --- radix-partitioned build (qh_partition.hip) -------------------------------------
// destination of one partition level
struct qh_part_dst {
  uint64_t* out;                   // buckets, out_cap items each
  uint64_t out_cap;
  unsigned long long* out_counts;  // fill count per bucket (zeroed by the caller)
  int bin_shift, bin_bits;         // bucket of an item = (hash >> bin_shift) & (2^bin_bits - 1)
  int out_bin_shift;               // output bucket = (input bucket << out_bin_shift) | bucket
  uint64_t* spill;                 // items beyond a bucket's capacity (meta->spill_count)
  uint64_t spill_cap;
  // Distance in words between the counters of consecutive buckets (0 = 1). Every tile of every
  // workgroup reserves bucket space with one returning atomic per bucket; counters that share a
  // cache line serialise in one L2 channel (measured: 0.5 ms for 64 adjacent counters), so the
  // first level keeps them QH_COUNT_STRIDE words apart.
  uint32_t out_count_stride;
  // > 0: the buckets hold PACKED items, 32 bits each: {remainder | position in the window : QH_WINDOW_BITS | quality : 1}
  // -- what a window kernel needs of an item once the bucket names the window (qh_items_pack_gsize; the value is the
  // table's lsize + shard_bits, the shift that yields the remainder). Items that do not fit a bucket are spilled whole.
  int32_t pack_gsize;
};
// words of scratch that qh_launch_index_fastq / qh_launch_segments use behind the (n + 1) counts they scan
#define QH_SCAN_SCRATCH 1024u
#define QH_COUNT_STRIDE 513u
// Window buckets can hold packed items when remainder, position and quality bit fit 32 bits (k = 24 from 2^30 slots on,
// shorter mers earlier): returns the gsize to pack with, or 0. The buckets are the bulk of the partitioned build's
// traffic at large tables: level 2 writes them, the window pass reads them.
static inline int32_t qh_items_pack_gsize(const qh_geom& g) {
  return g.rbits + QH_WINDOW_BITS + 1 <= 32 && g.lsize >= QH_WINDOW_BITS ? g.lsize + g.shard_bits : 0;
}
// Per-chunk device arrays of the FASTQ source (qh_partition.hip):
//   tile_sums  ceil(rec_cap/1024) + 1 words     scan scratch; ends with the segment total
//   seg_desc   max_segs x 8 B                    {first base offset, (distance to its quality) << 4 | k-mers}
//   planes     4 planes x plane_words words      v, h, l0, l1: one bit per byte of the chunk;
//                                                plane_words = 256 per 16 KiB tile of the chunk + 2 (even)
#define QH_SEG_KMERS 12
struct qh_fastq_aux {
  uint32_t* tile_sums;
  uint2* seg_desc;
  uint64_t* planes;
  uint64_t plane_words;
};
// chunk text -> bit planes (one streaming pass, no record knowledge)
void qh_launch_text_planes(const uint8_t* buf, size_t len, const qh_geom& g, const qh_fastq_aux& a, hipStream_t s);
// record table -> segment descriptors (3 small kernels)
void qh_launch_segments(const uint4* recs, uint32_t rec_cap, qh_meta* meta, const qh_geom& g,
                        const qh_fastq_aux& a, void* zero_a, size_t zero_a_bytes, void* zero_b, size_t zero_b_bytes,
                        hipStream_t s);
void qh_launch_partition_fastq(const qh_fastq_aux& a, uint32_t rec_cap, qh_meta* meta, const qh_geom& g,
                               const qh_part_dst& d, int check_owner, hipStream_t s);
void qh_launch_partition_items(const uint64_t* in, uint64_t in_cap, const unsigned long long* in_counts,
                               uint32_t in_count_stride, uint32_t n_in_buckets, uint64_t in_n, int in_hashed, qh_meta* meta, const qh_geom& g,
                               const qh_part_dst& d, int check_owner, int count_instances, hipStream_t s);
// table_state: bit 0 = the table holds entries (load each window first), bit 1 = the table has not
// been cleared yet (windows without items must be zeroed here)
// A window whose items do not all find a slot within the reprobe limit is left as it was: fail[w] = 1 (0 for every
// other window) and meta->win_failed counts them; qh_launch_window_redo runs those buckets again on the grown table.
// pack_gsize: 0 = 64-bit items, else the buckets hold packed items made with this gsize (qh_part_dst::pack_gsize; for
// the redo it is the gsize the buckets were made with, not the grown table's)
void qh_launch_window_build(uint64_t* table, const qh_geom& g, const uint64_t* items, uint64_t cap,
                            const unsigned long long* counts, uint32_t n_windows, int table_state, uint32_t* fail,
                            qh_meta* meta, hipStream_t s, int pack_gsize = 0);
// n_failed: how many flags of prev_fail are set (the pass counted them); list: n_failed + 1 words of scratch
void qh_launch_window_redo(uint64_t* table, const qh_geom& g, const uint64_t* items, uint64_t cap,
                           const unsigned long long* counts, uint32_t n_buckets, uint32_t n_windows, uint32_t* fail,
                           const uint32_t* prev_fail, uint32_t n_prev, uint32_t n_failed, uint32_t* list, qh_meta* meta,
                           hipStream_t s, int pack_gsize = 0);

// largest bucket capacity for which qh_launch_window_build still takes its fastest counting kernel at this geometry
// (counters must stay above a bucket's fill); 0 = the capacity does not change the kernel
uint64_t qh_window_count_cap_limit(const qh_geom& g);

// --- read side ------------------------------------------------------------------------
void qh_launch_lookup(const uint64_t* table, const qh_geom& g, const uint64_t* mers, uint64_t n,
                      uint64_t* out, int get_val, hipStream_t s);
// out may be null (membership only); the two tables share k
void qh_launch_lookup_contaminant(const uint64_t* table, const qh_geom& g, const uint64_t* ctable, const qh_geom& cg,
                                  const uint64_t* mers, uint64_t n, uint64_t* out, uint8_t* flags, hipStream_t s);
// (qh_correct.hip: one walk for the four siblings where the geometry allows it, else the generic kernel)
void qh_launch_best_alternatives(const uint64_t* table, const qh_geom& g, const uint64_t* fmers,
                                 const uint8_t* backward, uint64_t n, uint64_t* counts,
                                 int32_t* ucode, int32_t* level, int32_t* nkept, hipStream_t s);
void qh_launch_best_alternatives_generic(const uint64_t* table, const qh_geom& g, const uint64_t* fmers,
                                         const uint8_t* backward, uint64_t n, uint64_t* counts,
                                         int32_t* ucode, int32_t* level, int32_t* nkept, hipStream_t s);
void qh_launch_stats(const uint64_t* table, const qh_geom& g, qh_meta* meta, hipStream_t s);
void qh_launch_histo(const uint64_t* table, const qh_geom& g, unsigned long long* histo, hipStream_t s);
void qh_launch_dump(const uint64_t* table, const qh_geom& g, uint64_t* mers, uint64_t* vals,
                    uint64_t cap, qh_meta* meta, hipStream_t s);

// values image words [w0, w0 + nwords) of the database file for this shard's slots
void qh_launch_vals_image(const uint64_t* table, const qh_geom& g, uint64_t first, uint64_t slots, uint64_t w0,
                          uint64_t nwords, uint64_t* out, hipStream_t s);

// wide form: the value fields of a file's values image (bits+1 wide, atomic_bits_array layout) -> side array
void qh_launch_vals_unpack(const uint64_t* image, int value_bits, uint64_t slots, uint64_t* wvals, hipStream_t s);

// --- roofline microbenchmarks -------------------------------------------------------
void qh_launch_random_access(uint64_t* table, int lsize, uint64_t n, int mode, uint64_t seed,
                             unsigned long long* sink, hipStream_t s);
void qh_launch_lds_rate(int mode, int iters, int blocks, unsigned long long* sink, hipStream_t s);
