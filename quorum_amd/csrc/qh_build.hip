// qh_build.hip -- build side of the C ABI (include/quorum_hip.h): the host logic
// around the gfx950 kernels that replaces hash_with_quality + quality_mer_counter
// (src/mer_database.hpp:65-188, src/create_database.cc:44-96).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <errno.h>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "qh_host.h"
#include "qh_kernels.h"
#include "qh_dbfile.h"
#include "qh_internal.h"

// ------------------------------------------------------------------- errors

static thread_local std::string g_last_error;

void qh_set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

int qh_fail(int status, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return status;
}

extern "C" const char* qh_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* qh_version(void) { return "quorum_hip 0.1 (gfx950)"; }

// ------------------------------------------- scratch of the no-allocation entry point
//
// qh_build_add_mers_device_noalloc (end of this file) runs k_insert_items with its counters and its
// overflow queue in this file-scope array instead of the handle's hipMalloc'ed d_meta / d_ovf.
// Layout: [qh_meta, padded to 256 bytes][QH_NOALLOC_QUEUE_ITEMS items]. Every GPU holds its own copy
// at its own address: the array's NAME is only a host-side stand-in, host code uses the address
// hipGetSymbolAddress returned for the device (g_scratch_ptr), resolved once per device in qh_build_create.
#define QH_NOALLOC_MAX_ITEMS (1ull << 24)   // items per call
#define QH_NOALLOC_QUEUE_ITEMS (1u << 16)   // overflow queue: 512 KiB
#define QH_NOALLOC_META_BYTES 256u
#define QH_NOALLOC_SCRATCH_BYTES (QH_NOALLOC_META_BYTES + 8u * QH_NOALLOC_QUEUE_ITEMS)
static_assert(sizeof(qh_meta) <= QH_NOALLOC_META_BYTES, "qh_meta must fit the head of the scratch array");
static __device__ __align__(256) unsigned char scratch[QH_NOALLOC_SCRATCH_BYTES];

#define QH_MAX_DEVICES 64
static std::mutex g_scratch_mu;
static void* g_scratch_ptr[QH_MAX_DEVICES];  // device address of `scratch` on each device, 0 = not resolved yet

// with `device` current: the address of this device's copy of the scratch array
static int scratch_resolve(int device, void** out) {
  if (device < 0 || device >= QH_MAX_DEVICES) return qh_fail(QH_EBADARG, "device id %d out of range", device);
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_scratch_ptr[device]) {
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) {
      (void)hipGetLastError();
      return qh_fail(QH_EHIP, "hipGetSymbolAddress(scratch) failed on device %d: %s", device, hipGetErrorString(e));
    }
    g_scratch_ptr[device] = p;
  }
  *out = g_scratch_ptr[device];
  return QH_OK;
}

// ----------------------------------------------------------------- geometry

static int bitsize(uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }

bool qh_make_geom(int k, int bits, uint64_t size, int min_qual, int reprobe_limit, int shard_bits,
                  int shard_id, int wbits_override, qh_geom& g, std::string& err) {
  char m[256];
  if (k < 1 || k > 31) { snprintf(m, sizeof m, "Mer length must be between 1 and 31 (got %d)", k); err = m; return false; }
  if (bits < 1 || bits > 63) { err = "The number of bits should be between 1 and 63"; return false; }  // create_database.cc:111-112
  if (size == 0) { err = "Initial hash size must be positive"; return false; }
  if (shard_bits < 0 || shard_bits > 8 || shard_id < 0 || shard_id >= (1 << shard_bits)) { err = "invalid shard specification"; return false; }
  if (reprobe_limit < 0) reprobe_limit = 0;
  int lsize = 0;
  while (lsize < 48 && (1ull << lsize) < size) ++lsize;  // large_hash::array: size -> 2^ceilLog2(size)
  if ((1ull << lsize) < size) { err = "hash size too large"; return false; }
  if (lsize < 4) lsize = 4;
  memset(&g, 0, sizeof g);
  g.k = k; g.bits = bits; g.vbits = bits + 1;
  g.lsize = lsize; g.shard_bits = shard_bits; g.shard_id = shard_id;
  g.wbits = wbits_override >= 0 ? wbits_override : (lsize < QH_WINDOW_BITS ? lsize : QH_WINDOW_BITS);
  if (g.wbits > lsize) g.wbits = lsize;
  uint64_t maxr = (1ull << g.wbits) - 1;
  g.max_reprobe = (int)((uint64_t)reprobe_limit < maxr ? (uint64_t)reprobe_limit : maxr);
  g.pbits = bitsize((uint64_t)g.max_reprobe + 1);
  const int gsize = lsize + shard_bits;
  g.rbits = 2 * k + 1 > gsize ? 2 * k + 1 - gsize : 0;  // the hash is one bit wider than the key (qh_spec.h)
  g.min_qual = min_qual;
  g.kmask = qh_low_mask(2 * k);
  g.hmask = qh_low_mask(2 * k + 1);
  g.vmask = qh_low_mask(g.vbits);
  g.max_val = ~0ull >> (64 - bits);  // mer_database.hpp:83
  if (g.vbits + g.pbits + g.rbits > 64) {
    // upstream keeps the values in an array of their own and takes -b up to 63 at any -m / -s
    // (create_database.cc:111-112, mer_database.hpp:78-83). Where key, probe and value do not share one 64-bit word,
    // so does this table: the wide form (qh_spec.h), a compatibility route on the direct global-CAS path.
    if (g.pbits + g.rbits > 64) { err = "unsupported geometry: remainder and probe fields exceed one 64-bit slot"; return false; }
    g.wide = 1;
    g.vbits = 0;
    g.vmask = 0;
  }
  return true;
}

// --------------------------------------------------------------- the handle

static int ensure(void** p, size_t* cap, size_t need, const char* what) {
  if (*cap >= need && *p) return QH_OK;
  if (*p) { hipError_t e = hipFree(*p); (void)e; *p = nullptr; *cap = 0; }
  size_t want = need + need / 8 + 256;
  hipError_t e = hipMalloc(p, want);
  if (e != hipSuccess) { *p = nullptr; return qh_fail(QH_ENOMEM, "hipMalloc of %zu bytes for %s failed: %s", want, what, hipGetErrorString(e)); }
  *cap = want;
  return QH_OK;
}

extern "C" int qh_build_create(qh_build** out, int k, int bits, uint64_t size, int min_qual, int reprobe_limit,
                               int device, int shard_bits, int shard_id) {
  if (!out) return qh_fail(QH_EBADARG, "null output handle");
  *out = nullptr;
  qh_geom g;
  std::string err;
  // sharded: `size` is the TOTAL table size; every shard gets size / 2^shard_bits
  if (shard_bits < 0 || shard_bits > 8) return qh_fail(QH_EBADARG, "invalid shard specification");
  uint64_t local = size >> shard_bits;
  if (shard_bits && (local << shard_bits) < size) ++local;
  if (!qh_make_geom(k, bits, local, min_qual, reprobe_limit, shard_bits, shard_id, -1, g, err))
    return qh_fail(QH_EBADARG, "%s", err.c_str());
  QH_HIP(hipSetDevice(device));
  qh_build* h = new qh_build();
  h->device = device;
  h->g = g;
  h->reprobe_limit = reprobe_limit;
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete h; return qh_fail(QH_EHIP, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
  h->own_stream = h->stream;
  e = hipStreamCreateWithFlags(&h->pstream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->prep_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->main_mark, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_pmeta, sizeof(qh_meta));
  if (e != hipSuccess) { int rc = qh_fail(QH_EHIP, "setting up the preparation stream failed: %s", hipGetErrorString(e)); qh_build_destroy(h); return rc; }
  { int src = scratch_resolve(device, &h->noalloc_scratch); if (src) { qh_build_destroy(h); return src; } }
  const size_t bytes = (size_t)8 << g.lsize;
  e = hipMalloc((void**)&h->table, bytes);
  if (e == hipSuccess && g.wide) e = hipMalloc((void**)&h->g.wvals, bytes);  // the side array of the wide form
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_meta, sizeof(qh_meta));
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_meta, sizeof(qh_meta), hipHostMallocDefault);
  if (e == hipSuccess) e = hipMemsetAsync(h->table, 0, bytes, h->stream);
  if (e == hipSuccess && g.wide) e = hipMemsetAsync(h->g.wvals, 0, bytes, h->stream);
  if (e == hipSuccess) e = hipMemsetAsync(h->d_meta, 0, sizeof(qh_meta), h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) {
    int rc = qh_fail(e == hipErrorOutOfMemory ? QH_ENOMEM : QH_EHIP, "allocating a %zu-byte table failed: %s", bytes, hipGetErrorString(e));
    qh_build_destroy(h);
    return rc;
  }
  memset(&h->st, 0, sizeof h->st);
  h->st.slots = 1ull << g.lsize;
  if (const char* m = getenv("QH_BUILD_MODE")) h->mode = !strcmp(m, "direct") ? 1 : !strcmp(m, "partition") ? 2 : 0;
  if (const char* b = getenv("QH_LEVEL1_BITS")) h->level1_bits = atoi(b);
  if (const char* d = getenv("QH_DEFER_BYTES")) { const long long v = atoll(d); h->defer_budget = v < 0 ? -1 : v; }
  *out = h;
  return QH_OK;
}

extern "C" void qh_build_destroy(qh_build* h) {
  if (!h) return;
  hipError_t e = hipSetDevice(h->device); (void)e;
  if (h->stream) { e = hipStreamSynchronize(h->stream); (void)e; }
  if (h->pstream) { e = hipStreamSynchronize(h->pstream); (void)e; e = hipStreamDestroy(h->pstream); (void)e; }
  if (h->prep_done) { e = hipEventDestroy(h->prep_done); (void)e; }
  if (h->main_mark) { e = hipEventDestroy(h->main_mark); (void)e; }
  void* alts[] = {h->alt.recs, h->alt.tiles, h->alt.nlplane, h->alt.rbsums, h->alt.segdesc, h->alt.planes, h->d_pmeta};
  for (void* p : alts) if (p) { e = hipFree(p); (void)e; }
  void* ptrs[] = {h->table, h->g.wvals, h->d_meta, h->d_chunk, h->d_recs, h->d_tiles, h->d_ovf[0], h->d_ovf[1], h->d_items, h->d_shard_counts, h->d_rbsums, h->d_segdesc, h->d_planes, h->d_nlplane,
                  h->d_bins[0], h->d_bins[1], h->d_bin_counts[0], h->d_bin_counts[1], h->d_wfail[0], h->d_wfail[1], h->d_bspill, h->d_wlist};
  for (void* p : ptrs) if (p) { e = hipFree(p); (void)e; }
  if (h->h_meta) { e = hipHostFree(h->h_meta); (void)e; }
  for (auto& pr : h->evk) for (hipEvent_t ev : pr) if (ev) { e = hipEventDestroy(ev); (void)e; }
  for (auto& pr : h->evp) for (hipEvent_t ev : pr) if (ev) { e = hipEventDestroy(ev); (void)e; }
  if (h->own_stream) { e = hipStreamDestroy(h->own_stream); (void)e; }
  delete h;
}

extern "C" int qh_build_set_stream(qh_build* h, void* hip_stream) {
  if (!h) return qh_fail(QH_EBADARG, "null handle");
  int rc = qh_build_sync(h);
  h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
  return rc;
}

// -------------------------------------------------------------- growth / sync

// qh_build_reset defers the clearing of the table: a partitioned chunk writes every window anyway.
// Everything else that reads or updates the table clears it first.
static int materialize(qh_build* h) {
  if (!h->zero_pending) return QH_OK;
  QH_HIP(hipSetDevice(h->device));
  QH_HIP(hipMemsetAsync(h->table, 0, (size_t)8 << h->g.lsize, h->stream));
  if (h->g.wide) QH_HIP(hipMemsetAsync(h->g.wvals, 0, (size_t)8 << h->g.lsize, h->stream));
  h->zero_pending = false;
  return QH_OK;
}

// the table-doubling half of handle_full_ary (mer_database.hpp:142-184): allocate a
// table of at least 2^target_lsize slots, re-insert every key carrying its value
static int grow_table(qh_build* h, int target_lsize) {
  int mrc = materialize(h);
  if (mrc) return mrc;
  uint64_t* new_table = nullptr;
  qh_geom ng;
  for (int lsize = target_lsize;; ++lsize) {
    std::string err;
    if (lsize > 44 || !qh_make_geom(h->g.k, h->g.bits, 1ull << lsize, h->g.min_qual, h->reprobe_limit,
                                    h->g.shard_bits, h->g.shard_id, -1, ng, err))
      return qh_fail(QH_EFULL, "Hash is full");
    const size_t bytes = (size_t)8 << ng.lsize;
    hipError_t e = hipMalloc((void**)&new_table, bytes);
    if (e == hipSuccess && ng.wide) {  // new_vals_ (mer_database.hpp:148)
      e = hipMalloc((void**)&ng.wvals, bytes);
      if (e != hipSuccess) { (void)hipFree(new_table); new_table = nullptr; }
    }
    if (e != hipSuccess) {  // new_keys_ allocation threw: FULL (mer_database.hpp:149-163)
      (void)hipGetLastError();
      return qh_fail(QH_EFULL, "Hash is full (cannot allocate a %zu-byte table: %s)", bytes, hipGetErrorString(e));
    }
    QH_HIP(hipMemsetAsync(new_table, 0, bytes, h->stream));
    if (ng.wide) QH_HIP(hipMemsetAsync(ng.wvals, 0, bytes, h->stream));
    QH_HIP(hipMemsetAsync(h->d_meta, 0, sizeof(qh_meta), h->stream));
    qh_launch_rehash(h->table, h->g, new_table, ng, h->d_meta, h->stream);
    QH_HIP(hipGetLastError());
    QH_HIP(hipMemcpyAsync(h->h_meta, h->d_meta, sizeof(qh_meta), hipMemcpyDeviceToHost, h->stream));
    QH_HIP(hipStreamSynchronize(h->stream));
    if (!h->h_meta->rehash_fail) break;
    QH_HIP(hipFree(new_table));  // a key did not fit: try a bigger table
    new_table = nullptr;
    if (ng.wvals) QH_HIP(hipFree(ng.wvals));
  }
  QH_HIP(hipFree(h->table));
  if (h->g.wvals) QH_HIP(hipFree(h->g.wvals));
  h->table = new_table;
  h->g = ng;
  h->st.slots = 1ull << ng.lsize;
  h->st.grows += 1;
  return QH_OK;
}

// What a chunk or a window pass could not place: items that hit the reprobe limit on the direct path wait in the
// overflow list, windows that an item overflowed were left as they were with their buckets intact (k_window_build).
// The table doubles (handle_full_ary, mer_database.hpp:137-187) and both are run again, until nothing is left.
static int resolve_full(qh_build* h, uint64_t n_ovf, uint32_t n_winfail, uint32_t n_winhot) {
  uint32_t n_prev = h->defer_nw;  // windows of the pass whose flags are in d_wfail[wf]
  int wf = 0;
  while (n_ovf || n_winfail || n_winhot) {
    const int src = h->cur_ovf, dst = src ^ 1;
    int rc;
    if (n_ovf && (rc = ensure((void**)&h->d_ovf[dst], &h->ovf_cap[dst], n_ovf * 8, "overflow list"))) return rc;
    // windows that only met a hot key (watched counters) run again at this table size, with the compare-and-swap
    // kernel; what is full grows first
    if ((n_ovf || n_winfail) && (rc = grow_table(h, h->g.lsize + 1))) return rc;
    n_winfail += n_winhot;
    n_winhot = 0;
    QH_HIP(hipMemsetAsync(h->d_meta, 0, sizeof(qh_meta), h->stream));
    if (n_ovf)
      qh_launch_insert_items(h->d_ovf[src], n_ovf, nullptr, 0, 1, h->d_meta, h->g, h->table, h->d_ovf[dst], h->ovf_cap[dst] / 8, h->stream);
    uint32_t nw = 0;
    if (n_winfail) {
      nw = 1u << (h->g.lsize - QH_WINDOW_BITS);
      if ((rc = ensure((void**)&h->d_wfail[wf ^ 1], &h->wfail_cap[wf ^ 1], (size_t)nw * 4, "window flags"))) return rc;
      if ((rc = ensure((void**)&h->d_wlist, &h->wlist_cap, ((size_t)n_winfail + 1) * 4, "failed windows"))) return rc;
      qh_launch_window_redo(h->table, h->g, h->d_bins[1], h->defer_capw, h->d_bin_counts[1], h->defer_nw, nw,
                            h->d_wfail[wf ^ 1], h->d_wfail[wf], n_prev, n_winfail, h->d_wlist, h->d_meta, h->stream, h->defer_pack);
      h->st.window_passes += 1;
    }
    QH_HIP(hipGetLastError());
    QH_HIP(hipMemcpyAsync(h->h_meta, h->d_meta, sizeof(qh_meta), hipMemcpyDeviceToHost, h->stream));
    QH_HIP(hipStreamSynchronize(h->stream));
    if (h->h_meta->ovf_lost) return qh_fail(QH_EFULL, "Hash is full (overflow list exhausted)");
    h->st.distinct += h->h_meta->new_keys;
    if (n_ovf) h->cur_ovf = dst;
    n_ovf = h->h_meta->ovf_count;
    if (n_winfail) { wf ^= 1; n_prev = nw; }
    n_winfail = h->h_meta->win_failed;
  }
  return QH_OK;
}

extern "C" int qh_build_grow_to(qh_build* h, int lsize) {
  if (!h) return qh_fail(QH_EBADARG, "null handle");
  int rc = qh_build_sync(h);
  if (rc) return rc;
  QH_HIP(hipSetDevice(h->device));
  if (h->g.lsize >= lsize) return QH_OK;
  return grow_table(h, lsize);
}

static int launch_flush(qh_build* h, bool same_epoch);

// Collects the chunk in flight (counters, deferred errors, growth). flush: the public meaning of qh_build_sync --
// everything fed so far is in the table; without it a batch of deferred chunks stays open (begin_chunk).
// the segment kernels of early preparations that have ended (one may still run: that of the chunk being handed over)
static void fold_prep_timing(qh_build* h) {
  for (int i = 0; i < 2; ++i) {
    if (!h->evp_pending[i]) continue;
    if (hipEventQuery(h->evp[i][1]) != hipSuccess) { (void)hipGetLastError(); continue; }
    float ms = 0;
    if (hipEventElapsedTime(&ms, h->evp[i][0], h->evp[i][1]) == hipSuccess) { h->kernel_ms[0] += ms; h->kernel_launches[0] += 1; }
    h->evp_pending[i] = false;
  }
}

static void apply_reset(qh_build* h);

static int sync_core(qh_build* h, bool flush) {
  if (h->failed) return qh_fail(h->failed, "%s", h->failed_msg.c_str());
  for (;;) {
    if (flush && h->defer_open) { int frc = launch_flush(h, false); if (frc) return frc; }
    if (!h->pending) return QH_OK;
    QH_HIP(hipSetDevice(h->device));
    QH_HIP(hipMemcpyAsync(h->h_meta, h->d_meta, sizeof(qh_meta), hipMemcpyDeviceToHost, h->stream));
    QH_HIP(hipStreamSynchronize(h->stream));
    h->pending = false;
    for (int i = 0; i < 4; ++i) {
      if (!h->evk_pending[i]) continue;
      float ms = 0;
      if (hipEventElapsedTime(&ms, h->evk[i][0], h->evk[i][1]) == hipSuccess) { h->kernel_ms[i] += ms; h->kernel_launches[i] += 1; }
      h->evk_pending[i] = false;
    }
    fold_prep_timing(h);
    const qh_meta m = *h->h_meta;
    const bool was_deferred = h->pending_deferred;
    const size_t flen = h->pending_fastq_len;
    h->pending_deferred = false;
    h->pending_fastq_len = 0;
    if (m.format_err) {
      const char* why = (m.format_err & QH_FMT_LINES_MOD4)     ? "line count is not a multiple of 4"
                        : (m.format_err & QH_FMT_TOO_MANY)     ? "records shorter than the fast path supports"
                        : (m.format_err & QH_FMT_NO_AT)        ? "a record does not start with '@'"
                        : (m.format_err & QH_FMT_NO_PLUS)      ? "a separator line does not start with '+'"
                        : (m.format_err & QH_FMT_LEN_MISMATCH) ? "sequence and quality lengths differ"
                        : (m.format_err & QH_FMT_LINE_TOO_LONG) ? "a line of 256 MiB or more"
                                                               : "empty sequence";
      return qh_fail(QH_EFORMAT, "chunk is not strict 4-line FASTQ (%s, near record %u); nothing was inserted", why, m.first_bad);
    }
    h->st.instances += m.instances;
    h->st.hq_instances += m.hq_instances;
    h->st.distinct += m.new_keys;
    h->st.foreign += m.foreign;
    h->st.records += m.nrec;
    if (flen && m.instances) {
      const double r = (double)m.instances / (double)flen;
      if (r > h->items_per_byte) h->items_per_byte = r;
    }
    if (was_deferred) h->defer_items += m.instances;
    int rc = QH_OK;
    if (m.ovf_lost) rc = qh_fail(QH_EFULL, "Hash is full (overflow list exhausted)");
    else if (h->defer_open && m.spill_count) {
      // items of the last chunk that found their bucket full (windows differ by a few per cent in what they receive)
      // wait in the spill list, which the next chunk reuses: they move to the batch's own list; when that is full
      // the batch is applied now (the device-side spill count is kept, the counters just collected are not
      // counted twice)
      if ((h->bspill_n + m.spill_count) * 8 <= h->bspill_cap) {
        QH_HIP(hipMemcpyAsync(h->d_bspill + h->bspill_n, h->defer_spill, m.spill_count * 8, hipMemcpyDeviceToDevice, h->stream));
        QH_HIP(hipMemsetAsync(&h->d_meta->spill_count, 0, 8, h->stream));
        h->bspill_n += m.spill_count;
      } else if ((rc = launch_flush(h, false)) == QH_OK) {
        continue;
      }
    } else if (m.ovf_count || m.win_failed || m.win_hot) {
      if (h->defer_open) rc = qh_fail(QH_EHIP, "internal: overflow items while a deferred batch is open");
      else rc = resolve_full(h, m.ovf_count, m.win_failed, m.win_hot);
    }
    if (rc) { h->failed = rc; h->failed_msg = g_last_error; }
    return rc;
  }
}

// Collects the chunk in flight (flush: after the window pass over what waits in the buckets). A reset that was asked
// for while a chunk was in flight (qh_build_reset) is applied here, once that chunk has been collected: what waited in
// the buckets goes with the table.
static int sync_impl(qh_build* h, bool flush) {
  if (!h) return qh_fail(QH_EBADARG, "null handle");
  if (h->reset_pending) {
    h->reset_pending = false;
    const int rc = h->pending ? sync_core(h, false) : QH_OK;
    if (rc && rc != QH_EFORMAT) return rc;  // (what qh_build_reset itself reports when it collects)
    apply_reset(h);
  }
  return sync_core(h, flush);
}

extern "C" int qh_build_sync(qh_build* h) { return sync_impl(h, true); }

// HIP events around the main kernels of a chunk, on the stream they are launched on
// (which: 0 direct build / item insert, 1 partition level 1 incl. extraction,
//  2 partition level 2, 3 window build)
static void timing_begin(qh_build* h, int which = 0) {
  if (h->timing && h->evk[which][0]) { hipError_t e = hipEventRecord(h->evk[which][0], h->stream); (void)e; }
}
static void timing_end(qh_build* h, int which = 0) {
  if (h->timing && h->evk[which][1]) { hipError_t e = hipEventRecord(h->evk[which][1], h->stream); (void)e; h->evk_pending[which] = true; }
}

extern "C" int qh_build_enable_timing(qh_build* h, int on) {
  if (!h) return qh_fail(QH_EBADARG, "null handle");
  int rc = qh_build_sync(h);
  if (rc) return rc;
  QH_HIP(hipSetDevice(h->device));
  if (on && !h->evk[0][0]) {
    for (auto& pr : h->evk) for (hipEvent_t& ev : pr) QH_HIP(hipEventCreate(&ev));
    for (auto& pr : h->evp) for (hipEvent_t& ev : pr) QH_HIP(hipEventCreate(&ev));
  }
  h->timing = on != 0;
  h->evp_pending[0] = h->evp_pending[1] = false;
  for (int i = 0; i < 4; ++i) { h->kernel_ms[i] = 0; h->kernel_launches[i] = 0; h->evk_pending[i] = false; }
  return QH_OK;
}

extern "C" int qh_build_kernel_time(qh_build* h, int which, double* total_ms, uint64_t* launches) {
  if (!h || !total_ms || !launches || which < 0 || which > 3) return qh_fail(QH_EBADARG, "bad argument");
  int rc = qh_build_sync(h);
  *total_ms = h->kernel_ms[which]; *launches = h->kernel_launches[which];
  return rc;
}

extern "C" int qh_build_set_mode(qh_build* h, int mode, int level1_bits) {
  if (!h || mode < 0 || mode > 2 || level1_bits > 12) return qh_fail(QH_EBADARG, "bad argument");
  h->mode = mode;
  h->level1_bits = level1_bits;
  return QH_OK;
}

static bool use_partition(const qh_build* h, size_t max_items);

// upper estimate of the items a FASTQ chunk of `len` bytes brings: len / 2 (a base and a quality per item at least),
// or what chunks before it brought per byte, with a margin
static size_t est_fastq_items(const qh_build* h, size_t len) {
  const size_t ub = len / 2;
  if (h->items_per_byte <= 0) return ub;
  const size_t e = (size_t)((double)len * h->items_per_byte * 1.05) + 4096;
  return e < ub ? e : ub;
}

extern "C" int qh_build_set_defer(qh_build* h, long long max_bytes) {
  if (!h) return qh_fail(QH_EBADARG, "null handle");
  int rc = qh_build_sync(h);  // an open batch was sized under the old budget
  if (rc) return rc;
  h->defer_budget = max_bytes < 0 ? -1 : max_bytes;
  return QH_OK;
}

// Before a chunk is enqueued: the chunk in flight is collected; an open batch of deferred chunks is flushed when this
// chunk does not go into it (direct path) or does not fit it any more.
static int begin_chunk(qh_build* h, size_t max_items, size_t est_items) {
  int rc = sync_impl(h, false);
  if (rc) return rc;
  QH_HIP(hipSetDevice(h->device));
  if (h->defer_open) {
    const uint32_t nw_now = h->g.lsize >= QH_WINDOW_BITS ? 1u << (h->g.lsize - QH_WINDOW_BITS) : 0;
    if (!use_partition(h, max_items) || nw_now != h->defer_nw || h->defer_items + est_items > h->defer_cap_items) {
      if ((rc = launch_flush(h, false))) return rc;
      if ((rc = sync_impl(h, false))) return rc;
    }
  }
  rc = ensure((void**)&h->d_ovf[h->cur_ovf], &h->ovf_cap[h->cur_ovf], (max_items + 1) * 8, "overflow list");
  if (rc) return rc;
  QH_HIP(hipMemsetAsync(h->d_meta, 0, sizeof(qh_meta), h->stream));
  return QH_OK;
}

// ------------------------------------------------- radix-partitioned chunk path

// A chunk goes through the partitioned build when the table has enough 8192-slot
// windows to fill the chip and the chunk is large enough to amortise one read+write
// of the table (16 B/slot); small chunks into big tables stay on the direct path.
static bool use_partition(const qh_build* h, size_t max_items) {
  const int wid_bits = h->g.lsize - QH_WINDOW_BITS;
  if (h->g.wbits != QH_WINDOW_BITS || wid_bits < 0 || wid_bits > 24) return false;
  if (h->g.wide) return false;  // the LDS windows hold whole entries: the wide form stays on the direct path
  if (h->mode == 1) return false;
  if (h->mode == 2) return true;
  if (wid_bits < 8) return false;
  // a chunk pays for one pass over the table from about slots / 8 items on; with the window pass deferred over a
  // batch of chunks (defer_budget) the table is streamed once per batch, so smaller chunks join an open batch and
  // chunks from slots / 64 items on open one
  if (h->defer_budget != 0 && (h->defer_open || max_items >= (((size_t)1 << h->g.lsize) >> 6))) return true;
  return max_items >= (((size_t)1 << h->g.lsize) >> 3);
}

// device arrays of the FASTQ source for a chunk of buf_len bytes, at most rec_cap records / max_bases bases
static int fastq_aux(qh_build* h, size_t buf_len, size_t rec_cap, size_t max_bases, qh_fastq_aux* a) {
  const size_t segs = max_bases / QH_SEG_KMERS + rec_cap, plane_words = (buf_len + 16383) / 16384 * 256 + 2;  // whole 16 KiB tiles (k_index_text stores a tile's planes in 16-byte pieces), even
  int rc;
  if ((rc = ensure((void**)&h->d_rbsums, &h->rbsums_cap, (rec_cap / 1024 + 4 + QH_SCAN_SCRATCH) * 4, "segment sums"))) return rc;
  if ((rc = ensure((void**)&h->d_segdesc, &h->segdesc_cap, (segs + 2) * 8, "segment descriptors"))) return rc;
  if ((rc = ensure((void**)&h->d_planes, &h->planes_cap, plane_words * 32, "bit planes"))) return rc;
  *a = {h->d_rbsums, h->d_segdesc, h->d_planes, plane_words};
  return QH_OK;
}

struct part_source {
  const uint8_t* fastq = nullptr;    // FASTQ chunk whose record table is in h->d_recs
  size_t fastq_len = 0;              // bytes of that chunk
  bool planes_ready = false;         // the text planes were made by the indexing pass
  bool prepped = false;              // ... and the segment descriptors too, ahead of time (the bucket counters are cleared here)
  size_t rec_cap = 0;
  size_t max_bases = 0;              // upper bound of the chunk's bases
  const uint64_t* items = nullptr;   // or an item list
  size_t n_items = 0;
  int hashed = 0;
};

// bytes the window buckets of one batch may take (0: no deferral)
static size_t defer_budget_bytes(qh_build* h) {
  if (h->defer_budget == 0) return 0;
  if (h->defer_budget > 0) return (size_t)h->defer_budget;
  if (!h->free_at_first_use) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return 0; }
    h->free_at_first_use = free_b + h->bins_cap[1];
  }
  const size_t by_table = (size_t)24 << h->g.lsize;  // three times the table
  return by_table < h->free_at_first_use / 2 ? by_table : h->free_at_first_use / 2;
}

// The window pass over the open batch: every window is loaded (when the table holds entries), its bucket applied in
// LDS, and streamed back; then the items that found no room in a bucket or a window go in one by one.
// same_epoch: called at the end of the chunk that filled the buckets, whose counters are still being gathered in
// d_meta; otherwise the chunk was collected already and d_meta is cleared but for its spill count.
static int launch_flush(qh_build* h, bool same_epoch) {
  if (!h->defer_open) return QH_OK;
  QH_HIP(hipSetDevice(h->device));
  if (!same_epoch && !h->pending) {
    static_assert(offsetof(qh_meta, spill_count) == 32 && offsetof(qh_meta, instances) == 40, "qh_meta layout");
    QH_HIP(hipMemsetAsync(h->d_meta, 0, offsetof(qh_meta, spill_count), h->stream));
    QH_HIP(hipMemsetAsync(&h->d_meta->instances, 0, sizeof(qh_meta) - offsetof(qh_meta, instances), h->stream));
  }
  { int rc = ensure((void**)&h->d_wfail[0], &h->wfail_cap[0], (size_t)h->defer_nw * 4, "window flags"); if (rc) return rc; }
  if (h->bspill_n) {  // what the direct insertion below cannot place goes to the overflow list: room for all of it
    int rc = ensure((void**)&h->d_ovf[h->cur_ovf], &h->ovf_cap[h->cur_ovf], (h->bspill_n + h->defer_spill_cap + 1) * 8, "overflow list");
    if (rc) return rc;
  }
  timing_begin(h, 3);
  qh_launch_window_build(h->table, h->g, h->d_bins[1], h->defer_capw, h->d_bin_counts[1], h->defer_nw,
                         (h->table_dirty ? 1 : 0) | (h->zero_pending ? 2 : 0), h->d_wfail[0], h->d_meta, h->stream, h->defer_pack);
  timing_end(h, 3);
  // an early preparation (qh_build_add_fastq_device) waits for this pass to end: the pass keeps every CU's LDS, and
  // swarms of small workgroups from the other stream slipping in as its workgroups retire left it at half its
  // occupancy (measured: 17.8 instead of 10.7 ms for the 2^31-slot table)
  if (h->main_mark) QH_HIP(hipEventRecord(h->main_mark, h->stream));
  h->zero_pending = false;  // every window has been written
  // items that overflowed a bucket (the last chunk's, then the batch's list): plain global-CAS insertion
  qh_launch_insert_items(h->defer_spill, h->defer_spill_cap, reinterpret_cast<const unsigned long long*>(&h->d_meta->spill_count),
                         0, 1, h->d_meta, h->g, h->table, h->d_ovf[h->cur_ovf], h->ovf_cap[h->cur_ovf] / 8, h->stream);
  if (h->bspill_n)
    qh_launch_insert_items(h->d_bspill, h->bspill_n, nullptr, 0, 1, h->d_meta, h->g, h->table, h->d_ovf[h->cur_ovf],
                           h->ovf_cap[h->cur_ovf] / 8, h->stream);
  h->bspill_n = 0;
  QH_HIP(hipGetLastError());
  h->defer_open = false;
  h->defer_items = 0;
  h->table_dirty = true;
  h->pending = true;
  h->st.window_passes += 1;
  return QH_OK;
}

static int run_partitioned(qh_build* h, const part_source& src, size_t max_items, size_t est_items) {
  const qh_geom& g = h->g;
  const int wid_bits = g.lsize - QH_WINDOW_BITS;
  const uint32_t nw = 1u << wid_bits;
  // measured on MI355X (profiles/): two levels of about equal width beat one 4096-way level
  // (1.29 ms vs 1.97 ms for 152 M items); tables of up to 256 windows use one level
  // (level 1 loses more to a wide split than level 2: at 19 bits 9 + 10 takes 15.9 ms for 1.4 G items, 10 + 9 19.8 ms)
  // Level 1 pays more for every bit than level 2 (2^31 slots, 3.8 G items: level 1 16.6 / 18.7 / 23.5 / 33.7 ms at 7 / 8 / 9 /
  // 10 bits where its buckets lie badly, 17.4 / 18.5 at 8 / 9 where they lie well; level 2 15.7 / 16.4 /
  // 17.3 / 22.3 ms at 8 / 9 / 10 / 11): level 1 takes at most 8 bits as long as that leaves level 2 no more than 10
  int b1 = wid_bits <= 8 ? wid_bits : (wid_bits + 1) / 2;
  if (b1 > 8) b1 = wid_bits - 10 > 8 ? wid_bits - 10 : 8;
  if (b1 > 12) b1 = 12;
  if (h->level1_bits >= 0 && h->level1_bits <= wid_bits && wid_bits - h->level1_bits <= 12) b1 = h->level1_bits;
  const int b2 = wid_bits - b1;
  int rc;
  // Window buckets: those of the open batch, or a new batch. A batch is sized for several chunks when the budget
  // holds at least two of this size (the table is then streamed once for all of them, at the flush); otherwise
  // for this chunk alone, which is flushed at once.
  const bool open_new = !h->defer_open;
  bool batch = !open_new;
  size_t capw = h->defer_capw;
  // packed window buckets where the geometry allows (half the bytes for level 2 to write and the window pass to read)
  const bool no_pack = getenv("QH_NO_PACK") != nullptr;  // A/B runs and tests (read per chunk)
  const int pack = open_new ? (no_pack ? 0 : qh_items_pack_gsize(g)) : h->defer_pack;
  const size_t item_bytes = pack ? 4 : 8;
  if (open_new) {
    capw = max_items / nw + max_items / nw / 5 + 512;
    const size_t budget = defer_budget_bytes(h);
    if (budget) {
      size_t per_w = budget / item_bytes / nw;
      const uint64_t limit = qh_window_count_cap_limit(g);  // the counting window kernel needs counters above a bucket
      if (limit && per_w > limit) per_w = (size_t)limit;
      const size_t need_one = est_items / nw + est_items / nw / 5 + 512;
      if (per_w >= 2 * need_one && ensure((void**)&h->d_bins[1], &h->bins_cap[1], (size_t)nw * per_w * item_bytes, "window buckets") == QH_OK) {
        // (with the buckets a list of a sixteenth of their capacity for what overflows single buckets)
        if (ensure((void**)&h->d_bspill, &h->bspill_cap, ((size_t)nw * per_w / 16 + 65536) * 8, "batch spill list") == QH_OK) {
          capw = per_w;
          batch = true;
        }
      }
    }
    if (!batch && (rc = ensure((void**)&h->d_bins[1], &h->bins_cap[1], (size_t)nw * capw * item_bytes, "window buckets"))) return rc;
    h->defer_capw = capw;
    h->defer_pack = pack;
    h->defer_nw = nw;
    h->defer_items = 0;
    h->defer_cap_items = batch ? (uint64_t)nw * (capw - 512) / 6 * 5 : 0;  // windows differ: a fifth stays free
    h->bspill_n = 0;
  }
  const size_t cap1 = b2 ? max_items / ((size_t)1 << b1) + max_items / ((size_t)1 << b1) / 10 + 4096 : 0;
  // the counters are cleared by the segment kernel when there is one (FASTQ source), else by fills; the window
  // buckets' counters only when a batch is opened: later chunks append
  const size_t wc_bytes = ((size_t)nw * 8 + 15) & ~(size_t)15, l1c_bytes = b2 ? (((size_t)8 << b1) * QH_COUNT_STRIDE + 15) & ~(size_t)15 : 0;
  if (open_new && (rc = ensure((void**)&h->d_bin_counts[1], &h->bin_counts_cap[1], wc_bytes, "window bucket counts"))) return rc;
  if (b2) {
    if ((rc = ensure((void**)&h->d_bins[0], &h->bins_cap[0], ((size_t)cap1 << b1) * 8, "level-1 buckets"))) return rc;
    if ((rc = ensure((void**)&h->d_bin_counts[0], &h->bin_counts_cap[0], l1c_bytes, "level-1 bucket counts"))) return rc;
  }
  if (!src.fastq || src.prepped) {
    if (open_new) QH_HIP(hipMemsetAsync(h->d_bin_counts[1], 0, wc_bytes, h->stream));
    if (b2) QH_HIP(hipMemsetAsync(h->d_bin_counts[0], 0, l1c_bytes, h->stream));
  }
  // spill list = the idle overflow buffer; failures of its direct insertion go to the active one
  const int cur = h->cur_ovf, other = cur ^ 1;
  if ((rc = ensure((void**)&h->d_ovf[other], &h->ovf_cap[other], (max_items + 1) * 8, "spill list"))) return rc;
  uint64_t* spill = h->d_ovf[other];
  const uint64_t spill_cap = h->ovf_cap[other] / 8;
  h->defer_spill = spill;
  h->defer_spill_cap = spill_cap;

  qh_part_dst win_dst = {h->d_bins[1], capw, h->d_bin_counts[1], g.shard_bits + QH_WINDOW_BITS, b2 ? b2 : wid_bits,
                         b2, spill, spill_cap, 1, pack};
  qh_part_dst l1_dst = win_dst;
  if (b2) l1_dst = {h->d_bins[0], cap1, h->d_bin_counts[0], g.shard_bits + QH_WINDOW_BITS + b2, b1, 0, spill, spill_cap, QH_COUNT_STRIDE, 0};
  else win_dst.out_bin_shift = 0;
  const int check_owner = g.shard_bits > 0;

  if (src.fastq) {
    qh_fastq_aux aux;
    if ((rc = fastq_aux(h, src.fastq_len, src.rec_cap, src.max_bases, &aux))) return rc;
    if (!src.prepped) {
      timing_begin(h, 0);
      if (!src.planes_ready) qh_launch_text_planes(src.fastq, src.fastq_len, g, aux, h->stream);
      qh_launch_segments(h->d_recs, (uint32_t)src.rec_cap, h->d_meta, g, aux, open_new ? h->d_bin_counts[1] : nullptr,
                         open_new ? wc_bytes : 0, b2 ? h->d_bin_counts[0] : nullptr, l1c_bytes, h->stream);
      timing_end(h, 0);
    }
    // the next chunk's early preparation starts here at the earliest: beside this chunk's own preparation (the first
    // chunk of a run has one) both take as long as one after the other, and level 1 starts that much later
    if (h->main_mark) QH_HIP(hipEventRecord(h->main_mark, h->stream));
    timing_begin(h, 1);
    qh_launch_partition_fastq(aux, (uint32_t)src.rec_cap, h->d_meta, g, l1_dst, check_owner, h->stream);
  } else {
    timing_begin(h, 1);
    qh_launch_partition_items(src.items, src.n_items, nullptr, 1, 1, src.n_items, src.hashed, h->d_meta, g, l1_dst,
                              check_owner, 1, h->stream);
  }
  timing_end(h, 1);
  if (b2) {
    timing_begin(h, 2);
    qh_launch_partition_items(h->d_bins[0], cap1, h->d_bin_counts[0], QH_COUNT_STRIDE, 1u << b1, 0, 1, h->d_meta, g, win_dst, 0, 0, h->stream);
    timing_end(h, 2);
  }
  QH_HIP(hipGetLastError());
  h->defer_open = true;
  if (batch) {
    h->pending_deferred = true;  // its item count joins the batch's when the chunk is collected
    return QH_OK;
  }
  return launch_flush(h, true);
}

// ------------------------------------------------------------------ inputs

static const size_t kMaxChunk = 0xFFFFF000ull;  // 32-bit offsets inside a chunk

extern "C" int qh_build_add_fastq_device(qh_build* h, const void* dev_buf, size_t len) {
  if (!h) return qh_fail(QH_EBADARG, "null handle");
  if (len == 0) return QH_OK;
  if (!dev_buf || ((uintptr_t)dev_buf & 15)) return qh_fail(QH_EBADARG, "device chunk must be non-null and 16-byte aligned");
  if (len > kMaxChunk) return qh_fail(QH_EBADARG, "chunk larger than %zu bytes", kMaxChunk);
  const size_t est = est_fastq_items(h, len);
  const size_t ntiles = (len + QH_TILE_BYTES - 1) / QH_TILE_BYTES;
  const size_t rec_cap = len / 16 + 16;
  int rc;
  // Early preparation: while the previous chunk's split kernels still run on the handle's stream, this chunk's index
  // and segment kernels -- which need nothing of it -- fill the other set of per-chunk arrays on a stream of their
  // own; the host then collects the previous chunk as ever. Only when the handle runs on its own stream: a caller's
  // stream (qh_build_set_stream) orders the chunk's text behind whatever the caller enqueued there, which a second
  // stream would not see.
  const char* no_early = getenv("QH_NO_EARLY_PREP");  // (read per call: the tests switch it)
  const bool early = h->pending && h->last_partitioned && h->stream == h->own_stream && !h->failed && !no_early &&
                     use_partition(h, len / 2);
  if (early) {
    QH_HIP(hipSetDevice(h->device));
    std::swap(h->d_recs, h->alt.recs); std::swap(h->rec_cap, h->alt.rec_cap);
    std::swap(h->d_tiles, h->alt.tiles); std::swap(h->tile_cap, h->alt.tile_cap);
    std::swap(h->d_nlplane, h->alt.nlplane); std::swap(h->nlplane_cap, h->alt.nlplane_cap);
    std::swap(h->d_rbsums, h->alt.rbsums); std::swap(h->rbsums_cap, h->alt.rbsums_cap);
    std::swap(h->d_segdesc, h->alt.segdesc); std::swap(h->segdesc_cap, h->alt.segdesc_cap);
    std::swap(h->d_planes, h->alt.planes); std::swap(h->planes_cap, h->alt.planes_cap);
    qh_fastq_aux aux;
    if ((rc = ensure((void**)&h->d_tiles, &h->tile_cap, (ntiles + 1 + QH_SCAN_SCRATCH) * 4, "tile counts"))) return rc;
    if ((rc = ensure((void**)&h->d_recs, &h->rec_cap, rec_cap * 16, "record table"))) return rc;
    if ((rc = ensure((void**)&h->d_nlplane, &h->nlplane_cap, (len / 16 + 2) * 8 + 8192, "newline lists"))) return rc;
    if ((rc = fastq_aux(h, len, rec_cap, len / 2, &aux))) return rc;
    QH_HIP(hipStreamWaitEvent(h->pstream, h->main_mark, 0));
    QH_HIP(hipMemsetAsync(h->d_pmeta, 0, sizeof(qh_meta), h->pstream));
    qh_launch_index_fastq((const uint8_t*)dev_buf, len, h->d_tiles, h->d_nlplane, h->d_recs, (uint32_t)rec_cap, h->d_pmeta,
                          &aux, h->g.min_qual, h->pstream);
    // timed with event pairs of its own (that of qh_build_kernel_time(0) belongs to the chunk in flight), two in turn:
    // the previous early preparation may still be running beside its predecessor's last kernel, the one before has ended
    fold_prep_timing(h);
    const int ep = h->evp_cur ^= 1;
    const bool timed = h->timing && h->evp[ep][0] && !h->evp_pending[ep];
    if (timed) { hipError_t e = hipEventRecord(h->evp[ep][0], h->pstream); (void)e; }
    qh_launch_segments(h->d_recs, (uint32_t)rec_cap, h->d_pmeta, h->g, aux, nullptr, 0, nullptr, 0, h->pstream);
    if (timed) { hipError_t e = hipEventRecord(h->evp[ep][1], h->pstream); (void)e; h->evp_pending[ep] = true; }
    QH_HIP(hipGetLastError());
    QH_HIP(hipEventRecord(h->prep_done, h->pstream));
    rc = begin_chunk(h, len / 2, est);  // collects the previous chunk
    if (rc) { hipError_t e = hipStreamSynchronize(h->pstream); (void)e; return rc; }
    QH_HIP(hipStreamWaitEvent(h->stream, h->prep_done, 0));
    h->st.early_preps += 1;
    // lines, records and the format verdict: the first four words of the preparation's counters
    QH_HIP(hipMemcpyAsync(h->d_meta, h->d_pmeta, 16, hipMemcpyDeviceToDevice, h->stream));
  } else {
    if ((rc = begin_chunk(h, len / 2, est))) return rc;
    if ((rc = ensure((void**)&h->d_tiles, &h->tile_cap, (ntiles + 1 + QH_SCAN_SCRATCH) * 4, "tile counts"))) return rc;
    if ((rc = ensure((void**)&h->d_recs, &h->rec_cap, rec_cap * 16, "record table"))) return rc;
    if ((rc = ensure((void**)&h->d_nlplane, &h->nlplane_cap, (len / 16 + 2) * 8 + 8192, "newline lists"))) return rc;
  }
  const bool part = use_partition(h, len / 2);
  qh_fastq_aux aux;
  if (part && (rc = fastq_aux(h, len, rec_cap, len / 2, &aux))) return rc;
  // one pass over the text: newline plane for the record table and, for the partitioned build, the text planes
  if (!early)
    qh_launch_index_fastq((const uint8_t*)dev_buf, len, h->d_tiles, h->d_nlplane, h->d_recs, (uint32_t)rec_cap, h->d_meta,
                          part ? &aux : nullptr, h->g.min_qual, h->stream);
  if (part) {
    part_source src;
    src.fastq = (const uint8_t*)dev_buf;
    src.planes_ready = true;
    src.prepped = early;
    src.fastq_len = len;
    src.rec_cap = rec_cap;
    src.max_bases = len / 2;
    if ((rc = run_partitioned(h, src, len / 2, est))) return rc;
  } else {
    if ((rc = materialize(h))) return rc;
    timing_begin(h);
    qh_launch_build_direct((const uint8_t*)dev_buf, h->d_recs, h->d_meta, h->g, h->table, h->d_ovf[h->cur_ovf],
                           h->ovf_cap[h->cur_ovf] / 8, h->stream);
    timing_end(h);
    h->table_dirty = true;
  }
  QH_HIP(hipGetLastError());
  h->pending = true;
  h->pending_fastq_len = len;
  h->last_partitioned = part;
  return QH_OK;
}

static int stage_chunk(qh_build* h, const char* buf, size_t len, const char* extra, size_t extra_len, size_t* extra_off) {
  const size_t eo = (len + 15) & ~(size_t)15;
  int rc = ensure((void**)&h->d_chunk, &h->chunk_cap, eo + extra_len + 16, "chunk staging");
  if (rc) return rc;
  QH_HIP(hipMemcpyAsync(h->d_chunk, buf, len, hipMemcpyHostToDevice, h->stream));
  if (extra_len) QH_HIP(hipMemcpyAsync(h->d_chunk + eo, extra, extra_len, hipMemcpyHostToDevice, h->stream));
  if (extra_off) *extra_off = eo;
  return QH_OK;
}

extern "C" int qh_build_add_fastq(qh_build* h, const char* buf, size_t len) {
  if (!h) return qh_fail(QH_EBADARG, "null handle");
  if (len == 0) return QH_OK;
  if (!buf) return qh_fail(QH_EBADARG, "null chunk");
  if (len > kMaxChunk) return qh_fail(QH_EBADARG, "chunk larger than %zu bytes", kMaxChunk);
  int rc = sync_impl(h, false);  // the staging buffer is reused: the previous chunk must have been read (an open batch stays open)
  if (rc) return rc;
  QH_HIP(hipSetDevice(h->device));
  if ((rc = stage_chunk(h, buf, len, nullptr, 0, nullptr))) return rc;
  // pageable source: the copy must have consumed `buf` before we return
  QH_HIP(hipStreamSynchronize(h->stream));
  return qh_build_add_fastq_device(h, h->d_chunk, len);
}

static int submit_records(qh_build* h, const uint8_t* dev_buf, size_t buf_len, const qh_record* recs, size_t nrec,
                          size_t max_items) {
  int rc = begin_chunk(h, max_items, max_items);
  if (rc) return rc;
  if ((rc = ensure((void**)&h->d_recs, &h->rec_cap, (nrec + 1) * 16, "record table"))) return rc;
  static_assert(sizeof(qh_record) == 16, "qh_record must match the device record");
  QH_HIP(hipMemcpyAsync(h->d_recs, recs, nrec * 16, hipMemcpyHostToDevice, h->stream));
  const uint32_t n32 = (uint32_t)nrec;
  QH_HIP(hipMemcpyAsync(&h->d_meta->nrec, &n32, 4, hipMemcpyHostToDevice, h->stream));
  QH_HIP(hipStreamSynchronize(h->stream));  // recs / n32 are caller / stack memory
  if (use_partition(h, max_items)) {
    part_source src;
    src.fastq = dev_buf;
    src.fastq_len = buf_len;
    src.rec_cap = nrec;
    src.max_bases = max_items;
    if ((rc = run_partitioned(h, src, max_items, max_items))) return rc;
  } else {
    if ((rc = materialize(h))) return rc;
    timing_begin(h);
    qh_launch_build_direct(dev_buf, h->d_recs, h->d_meta, h->g, h->table, h->d_ovf[h->cur_ovf],
                           h->ovf_cap[h->cur_ovf] / 8, h->stream);
    timing_end(h);
    h->table_dirty = true;
  }
  QH_HIP(hipGetLastError());
  h->pending = true;
  return QH_OK;
}

extern "C" int qh_build_add_records_device(qh_build* h, const void* dev_buf, size_t len, const qh_record* recs, size_t nrec) {
  if (!h) return qh_fail(QH_EBADARG, "null handle");
  if (nrec == 0) return QH_OK;
  if (!dev_buf || !recs) return qh_fail(QH_EBADARG, "null buffer");
  if (len > kMaxChunk || nrec > 0xFFFFFFFFull) return qh_fail(QH_EBADARG, "chunk too large");
  size_t total = 0;
  for (size_t i = 0; i < nrec; ++i) {
    if ((size_t)recs[i].seq_off + recs[i].len > len || (size_t)recs[i].qual_off + recs[i].len > len)
      return qh_fail(QH_EBADARG, "record %zu lies outside the chunk", i);
    total += recs[i].len;
  }
  return submit_records(h, (const uint8_t*)dev_buf, len, recs, nrec, total);
}

// Sequential parser with jellyfish::whole_sequence_parser's FASTQ rules. Returns
// QH_OK and the records of every COMPLETE record; *consumed = bytes they span.
// Multi-line records are re-flowed into `scratch` (offsets relative to scratch_base).
int qh_parse_fastq_host(const char* buf, size_t len, bool final_chunk, std::vector<qh_record>& recs,
                        std::string& scratch, size_t* consumed, std::vector<qh_hdr>* hdrs) {
  size_t p = 0;
  *consumed = 0;
  auto eol = [&](size_t from) -> size_t {
    const void* e = from < len ? memchr(buf + from, '\n', len - from) : nullptr;
    return e ? (size_t)((const char*)e - buf) : len;
  };
  while (p < len) {
    if (buf[p] != '@')
      return p == 0 ? qh_fail(QH_EFORMAT, "Unsupported format: the chunk does not start with '@' (FASTA input has no qualities)")
                    : qh_fail(QH_EFORMAT, "Invalid fastq file: header missing");
    size_t e = eol(p);
    if (e >= len) { if (final_chunk) return qh_fail(QH_EFORMAT, "Truncated fastq file"); break; }
    const qh_hdr hd = {(uint32_t)(p + 1), (uint32_t)(e - p - 1)};  // is.ignore(1); getline(header)
    size_t q = e + 1;
    // sequence lines up to the '+' line
    const size_t seq_first = q;
    size_t slen = 0, nlines = 0, first_len = 0, spos = scratch.size();
    bool incomplete = false;
    for (;;) {
      if (q >= len) { incomplete = true; break; }
      if (buf[q] == '+') break;
      e = eol(q);
      if (e >= len && !final_chunk) { incomplete = true; break; }
      const size_t l = e - q;
      if (nlines == 0) first_len = l;
      if (nlines == 1) scratch.append(buf + seq_first, first_len);
      if (nlines >= 1) scratch.append(buf + q, l);
      slen += l; ++nlines;
      q = e < len ? e + 1 : len;
    }
    if (incomplete) { scratch.resize(spos); if (final_chunk) return qh_fail(QH_EFORMAT, "Truncated fastq file"); break; }
    const bool seq_multi = nlines > 1;
    // the '+' line
    e = eol(q);
    bool good = e < len;
    if (!good && !final_chunk) { scratch.resize(spos); break; }
    q = good ? e + 1 : len;
    // quality lines until qual.size() >= seq.size()
    const size_t qual_first = q, qspos = scratch.size();
    size_t qlen = 0, qlines = 0, qfirst = 0;
    incomplete = false;
    while (qlen < slen && good) {
      if (q >= len) { good = false; if (!final_chunk) incomplete = true; break; }
      e = eol(q);
      if (e >= len) { if (!final_chunk) { incomplete = true; break; } good = false; }
      const size_t l = e - q;
      if (qlines == 0) qfirst = l;
      if (qlines == 1) scratch.append(buf + qual_first, qfirst);
      if (qlines >= 1) scratch.append(buf + q, l);
      qlen += l; ++qlines;
      q = e < len ? e + 1 : len;
    }
    if (incomplete) { scratch.resize(spos); break; }
    if (qlen != slen) return qh_fail(QH_EFORMAT, "Invalid fastq file: wrong number of quals");
    if (q < len && buf[q] != '@') return qh_fail(QH_EFORMAT, "Invalid fastq file: header missing");
    if (slen > 0xFFFFFFFFull) return qh_fail(QH_EFORMAT, "read longer than 4 Gbases");
    const bool qual_multi = qlines > 1;
    qh_record r;
    r.len = (uint32_t)slen;
    // offsets into the scratch area are fixed up by the caller once its base is
    // known; reserved bit 0 / bit 1 mark a re-flowed sequence / quality
    r.seq_off = (uint32_t)(seq_multi ? spos : seq_first);
    r.qual_off = (uint32_t)(qual_multi ? qspos : qual_first);
    r.reserved = (seq_multi ? 1u : 0u) | (qual_multi ? 2u : 0u);
    recs.push_back(r);
    if (hdrs) hdrs->push_back(hd);
    p = q;
    *consumed = p;
  }
  return QH_OK;
}

extern "C" int qh_build_add_fastq_host_parsed(qh_build* h, const char* buf, size_t len, int final_chunk, size_t* consumed) {
  if (!h) return qh_fail(QH_EBADARG, "null handle");
  if (consumed) *consumed = 0;
  if (len == 0) return QH_OK;
  if (!buf) return qh_fail(QH_EBADARG, "null chunk");
  if (len > kMaxChunk / 2) return qh_fail(QH_EBADARG, "chunk larger than %zu bytes", kMaxChunk / 2);
  std::vector<qh_record> recs;
  std::string scratch;
  size_t used = 0;
  int rc = qh_parse_fastq_host(buf, len, final_chunk != 0, recs, scratch, &used, nullptr);
  if (rc) return rc;
  if (consumed) *consumed = used;
  if (recs.empty()) return QH_OK;
  if ((rc = sync_impl(h, false))) return rc;
  QH_HIP(hipSetDevice(h->device));
  size_t extra_off = 0;
  if ((rc = stage_chunk(h, buf, used, scratch.data(), scratch.size(), &extra_off))) return rc;
  size_t total = 0;
  for (size_t i = 0; i < recs.size(); ++i) {
    if (recs[i].reserved & 1u) recs[i].seq_off += (uint32_t)extra_off;
    if (recs[i].reserved & 2u) recs[i].qual_off += (uint32_t)extra_off;
    recs[i].reserved = 0;
    total += recs[i].len;
  }
  QH_HIP(hipStreamSynchronize(h->stream));
  return submit_records(h, h->d_chunk, extra_off + scratch.size() + 16, recs.data(), recs.size(), total);
}

// Contaminant set input: every canonical k-mer of a FASTA buffer, all with the
// quality bit set, through the same record path as FASTQ. FASTA as Jellyfish's
// whole_sequence_parser::read_fasta reads it: '>' header line, then sequence lines up to the
// next line that starts with '>'. Replaces `jellyfish count -C` (Makefile.am:54-55) + the
// binary_reader loop of contaminant_database (error_correct_reads.cc:86-90).
extern "C" int qh_build_add_fasta(qh_build* h, const char* buf, size_t len) {
  if (!h) return qh_fail(QH_EBADARG, "null handle");
  if (len == 0) return QH_OK;
  if (!buf) return qh_fail(QH_EBADARG, "null buffer");
  if (len > kMaxChunk / 2) return qh_fail(QH_EBADARG, "buffer larger than %zu bytes", kMaxChunk / 2);
  if (buf[0] != '>') return qh_fail(QH_EFORMAT, "Unsupported format: the FASTA buffer does not start with '>'");
  std::string seqs;
  std::vector<qh_record> recs;
  size_t p = 0, maxlen = 0, total = 0;
  auto eol = [&](size_t from) -> size_t {
    const void* e = from < len ? memchr(buf + from, '\n', len - from) : nullptr;
    return e ? (size_t)((const char*)e - buf) : len;
  };
  while (p < len) {
    p = eol(p);  // header line
    if (p < len) ++p;
    const size_t start = seqs.size();
    while (p < len && buf[p] != '>') {
      const size_t e = eol(p);
      seqs.append(buf + p, e - p);
      p = e < len ? e + 1 : e;
    }
    const size_t n = seqs.size() - start;
    if (n > 0xFFFFFFFFull) return qh_fail(QH_EFORMAT, "sequence longer than 4 Gbases");
    if (n) {
      qh_record r;
      r.seq_off = (uint32_t)start; r.len = (uint32_t)n; r.qual_off = 0; r.reserved = 0;
      recs.push_back(r);
      if (n > maxlen) maxlen = n;
      total += n;
    }
  }
  if (recs.empty()) return QH_OK;
  // one shared run of maximal qualities serves every record
  const std::string quals(maxlen, (char)127);
  int rc = sync_impl(h, false);
  if (rc) return rc;
  QH_HIP(hipSetDevice(h->device));
  size_t qoff = 0;
  if ((rc = stage_chunk(h, seqs.data(), seqs.size(), quals.data(), quals.size(), &qoff))) return rc;
  for (size_t i = 0; i < recs.size(); ++i) recs[i].qual_off = (uint32_t)qoff;
  QH_HIP(hipStreamSynchronize(h->stream));
  return submit_records(h, h->d_chunk, qoff + quals.size() + 16, recs.data(), recs.size(), total);
}

// The contaminant database as error_correct_reads.cc:692-708 opens it, built from FASTA at the
// run's k instead of read from a Jellyfish dump: a value-less set (bits = 1) whose membership
// test is "lookup != 0" (contaminant_database::is_contaminant, :94).
extern "C" int qh_contaminant_open_fasta(qh_query** out, const char* path, int k, int device) {
  if (!out || !path) return qh_fail(QH_EBADARG, "null argument");
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) return qh_fail(QH_EIO, "Failed to open contaminant file '%s'", path);
  std::string data;
  char tmp[1 << 16];
  size_t got;
  while ((got = fread(tmp, 1, sizeof tmp, f)) > 0) data.append(tmp, got);
  fclose(f);
  qh_build* b = nullptr;
  int rc = qh_build_create(&b, k, 1, data.size() + 1024, 0, 126, device, 0, 0);
  if (rc) return rc;
  if ((rc = qh_build_add_fasta(b, data.data(), data.size())) || (rc = qh_build_sync(b))) { qh_build_destroy(b); return rc; }
  rc = qh_query_from_table(b->device, b->table, b->g, out);
  if (rc == QH_OK) { (*out)->owns_table = true; b->table = nullptr; b->g.wvals = nullptr; }  // the set outlives its builder
  qh_build_destroy(b);
  return rc;
}

static int add_items_device(qh_build* h, const uint64_t* dev_items, size_t n, int hashed);

extern "C" int qh_build_add_mers_device(qh_build* h, const uint64_t* dev_items, size_t n) {
  return add_items_device(h, dev_items, n, 0);
}

extern "C" int qh_build_add_hashed_device(qh_build* h, const uint64_t* dev_items, size_t n) {
  return add_items_device(h, dev_items, n, 1);
}

static int add_items_device(qh_build* h, const uint64_t* dev_items, size_t n, int hashed) {
  if (!h) return qh_fail(QH_EBADARG, "null handle");
  if (n == 0) return QH_OK;
  if (!dev_items) return qh_fail(QH_EBADARG, "null items");
  int rc = begin_chunk(h, n, n);
  if (rc) return rc;
  if (use_partition(h, n)) {
    part_source src;
    src.items = dev_items;
    src.n_items = n;
    src.hashed = hashed;
    if ((rc = run_partitioned(h, src, n, n))) return rc;
  } else {
    if ((rc = materialize(h))) return rc;
    timing_begin(h);
    qh_launch_insert_items(dev_items, n, nullptr, 1, hashed, h->d_meta, h->g, h->table, h->d_ovf[h->cur_ovf], h->ovf_cap[h->cur_ovf] / 8, h->stream);
    timing_end(h);
    h->table_dirty = true;
  }
  QH_HIP(hipGetLastError());
  h->pending = true;
  return QH_OK;
}

extern "C" int qh_build_add_mers(qh_build* h, const uint64_t* mers, const uint8_t* quality, size_t n) {
  if (!h) return qh_fail(QH_EBADARG, "null handle");
  if (n == 0) return QH_OK;
  if (!mers || !quality) return qh_fail(QH_EBADARG, "null input");
  std::vector<uint64_t> items(n);
  for (size_t i = 0; i < n; ++i) {
    if (mers[i] & ~h->g.kmask) return qh_fail(QH_EBADARG, "mer %zu has bits above 2k", i);
    items[i] = (mers[i] << 1) | (quality[i] ? 1u : 0u);
  }
  int rc = sync_impl(h, false);
  if (rc) return rc;
  QH_HIP(hipSetDevice(h->device));
  if ((rc = ensure((void**)&h->d_items, &h->items_cap, n * 8, "item staging"))) return rc;
  QH_HIP(hipMemcpyAsync(h->d_items, items.data(), n * 8, hipMemcpyHostToDevice, h->stream));
  QH_HIP(hipStreamSynchronize(h->stream));
  return qh_build_add_mers_device(h, h->d_items, n);
}

extern "C" int qh_build_extract_partitioned_device(qh_build* h, const void* dev_buf, size_t len, uint64_t* dev_out,
                                                   uint64_t cap_per_shard, uint64_t* dev_counts) {
  if (!h) return qh_fail(QH_EBADARG, "null handle");
  if (!dev_out || !dev_counts) return qh_fail(QH_EBADARG, "null output");
  if (len && (!dev_buf || ((uintptr_t)dev_buf & 15))) return qh_fail(QH_EBADARG, "device chunk must be 16-byte aligned");
  if (len > kMaxChunk) return qh_fail(QH_EBADARG, "chunk larger than %zu bytes", kMaxChunk);
  int rc = qh_build_sync(h);
  if (rc) return rc;
  QH_HIP(hipSetDevice(h->device));
  QH_HIP(hipMemsetAsync(dev_counts, 0, sizeof(uint64_t) << h->g.shard_bits, h->stream));
  if (len == 0) return QH_OK;
  QH_HIP(hipMemsetAsync(h->d_meta, 0, sizeof(qh_meta), h->stream));
  const size_t ntiles = (len + QH_TILE_BYTES - 1) / QH_TILE_BYTES;
  const size_t rec_cap = len / 16 + 16;
  if ((rc = ensure((void**)&h->d_tiles, &h->tile_cap, (ntiles + 1 + QH_SCAN_SCRATCH) * 4, "tile counts"))) return rc;
  if ((rc = ensure((void**)&h->d_recs, &h->rec_cap, rec_cap * 16, "record table"))) return rc;
  if ((rc = ensure((void**)&h->d_nlplane, &h->nlplane_cap, (len / 16 + 2) * 8 + 8192, "newline lists"))) return rc;
  qh_fastq_aux aux;
  if ((rc = fastq_aux(h, len, rec_cap, len / 2, &aux))) return rc;
  qh_launch_index_fastq((const uint8_t*)dev_buf, len, h->d_tiles, h->d_nlplane, h->d_recs, (uint32_t)rec_cap, h->d_meta,
                        &aux, h->g.min_qual, h->stream);
  qh_launch_segments(h->d_recs, (uint32_t)rec_cap, h->d_meta, h->g, aux, nullptr, 0, nullptr, 0, h->stream);
  // bucket = owner shard = low shard_bits of the hash; a bucket that overflows is an error here
  qh_part_dst dst = {dev_out, cap_per_shard, (unsigned long long*)dev_counts, 0, h->g.shard_bits, 0, nullptr, 0, 1, 0};
  timing_begin(h, 1);
  qh_launch_partition_fastq(aux, (uint32_t)rec_cap, h->d_meta, h->g, dst, 0, h->stream);
  timing_end(h, 1);
  QH_HIP(hipGetLastError());
  // format / capacity verdict of this call
  QH_HIP(hipMemcpyAsync(h->h_meta, h->d_meta, sizeof(qh_meta), hipMemcpyDeviceToHost, h->stream));
  QH_HIP(hipStreamSynchronize(h->stream));
  if (h->h_meta->format_err) return qh_fail(QH_EFORMAT, "chunk is not strict 4-line FASTQ (flags 0x%x near record %u)", h->h_meta->format_err, h->h_meta->first_bad);
  if (h->h_meta->ovf_lost) return qh_fail(QH_EBADARG, "cap_per_shard too small for this chunk");
  h->st.records += h->h_meta->nrec;
  return QH_OK;
}

// The no-allocation variant of qh_build_add_mers_device (no counterpart in the reference): the same
// k_insert_items launch, with the kernel's counters (qh_meta) and overflow queue in this device's copy of
// the file-scope `scratch` array instead of buffers that ensure() would hipMalloc. Nothing here allocates:
// the table, the stream and the pinned meta mirror were made by qh_build_create. The call completes before
// it returns (the counters are read back), and a table that would have to grow is an error: growth allocates.
extern "C" int qh_build_add_mers_device_noalloc(qh_build* h, const uint64_t* dev_items, size_t n) {
  if (!h) return qh_fail(QH_EBADARG, "null handle");
  if (h->failed) return qh_fail(h->failed, "%s", h->failed_msg.c_str());
  if (n == 0) return QH_OK;
  if (!dev_items) return qh_fail(QH_EBADARG, "null items");
  if (n > QH_NOALLOC_MAX_ITEMS)
    return qh_fail(QH_EBADARG, "qh_build_add_mers_device_noalloc takes at most %llu items per call (got %zu)",
                   (unsigned long long)QH_NOALLOC_MAX_ITEMS, n);
  if (h->pending || h->defer_open)
    return qh_fail(QH_EBADARG, "a chunk of another entry point is in flight: call qh_build_sync first "
                               "(collecting it may have to grow the table, which allocates)");
  if (!h->noalloc_scratch) return qh_fail(QH_EHIP, "the scratch array was not resolved on device %d", h->device);
  QH_HIP(hipSetDevice(h->device));
  int rc = materialize(h);  // a memset of the existing table at most
  if (rc) return rc;
  // the device address from hipGetSymbolAddress, never the array's name
  unsigned char* base = static_cast<unsigned char*>(h->noalloc_scratch);
  qh_meta* meta = reinterpret_cast<qh_meta*>(base);
  uint64_t* queue = reinterpret_cast<uint64_t*>(base + QH_NOALLOC_META_BYTES);
  // the array keeps the last call's contents: the counters must start at zero (the queue is only read
  // below its fresh count)
  QH_HIP(hipMemsetAsync(meta, 0, QH_NOALLOC_META_BYTES, h->stream));
  qh_launch_insert_items(dev_items, n, nullptr, 1, 0, meta, h->g, h->table, queue, QH_NOALLOC_QUEUE_ITEMS, h->stream);
  QH_HIP(hipGetLastError());
  QH_HIP(hipMemcpyAsync(h->h_meta, meta, sizeof(qh_meta), hipMemcpyDeviceToHost, h->stream));
  QH_HIP(hipStreamSynchronize(h->stream));
  h->table_dirty = true;
  const qh_meta m = *h->h_meta;
  h->st.instances += m.instances;
  h->st.hq_instances += m.hq_instances;
  h->st.distinct += m.new_keys;
  h->st.foreign += m.foreign;
  if (m.ovf_lost || m.ovf_count) {  // the queued items are not in the table and growing it would allocate
    h->failed = qh_fail(QH_EFULL, "Hash is full (%llu items hit the reprobe limit; the no-allocation entry point "
                                  "cannot grow the table: create it large enough)",
                        (unsigned long long)m.ovf_count);
    h->failed_msg = g_last_error;
    return h->failed;
  }
  return QH_OK;
}

// -------------------------------------------------------------------- outputs

extern "C" int qh_build_get_stats(qh_build* h, qh_build_stats* stats) {
  if (!h || !stats) return qh_fail(QH_EBADARG, "null argument");
  int rc = qh_build_sync(h);
  *stats = h->st;
  return rc;
}

extern "C" int qh_build_finish(qh_build* h, qh_build_stats* stats) {
  if (!h) return qh_fail(QH_EBADARG, "null handle");
  int rc = qh_build_sync(h);
  if (stats) *stats = h->st;
  return rc;
}

static void apply_reset(qh_build* h) {
  h->zero_pending = true;  // cleared by the next partitioned chunk's window pass, or by materialize()
  const uint64_t slots = h->st.slots, grows = h->st.grows;
  memset(&h->st, 0, sizeof h->st);
  h->st.slots = slots; h->st.grows = grows;
  h->failed = 0;
  h->table_dirty = false;
  h->defer_open = false;  // what waited in the window buckets is dropped with the rest
  h->defer_items = 0;
  h->bspill_n = 0;
  h->pending_deferred = false;
}

// With a chunk or a window pass in flight on the handle's own stream the reset is only noted: it is applied when that
// work is collected, by whatever call comes next -- a FASTQ chunk handed over next is prepared beside it meanwhile
// (early preparation). An error of the work in flight that qh_build_reset would have returned is then returned by that
// next call.
extern "C" int qh_build_reset(qh_build* h) {
  if (!h) return qh_fail(QH_EBADARG, "null handle");
  if (h->pending && !h->failed && h->stream == h->own_stream && !getenv("QH_NO_EARLY_PREP")) {
    h->reset_pending = true;
    return QH_OK;
  }
  if (h->pending) { int rc = sync_impl(h, false); if (rc && rc != QH_EFORMAT) return rc; }
  h->reset_pending = false;
  apply_reset(h);
  return QH_OK;
}

// Starts the window pass over what waits in the buckets and returns; the next call that collects waits for it.
extern "C" int qh_build_flush(qh_build* h) {
  if (!h) return qh_fail(QH_EBADARG, "null handle");
  if (h->failed) return qh_fail(h->failed, "%s", h->failed_msg.c_str());
  if (h->reset_pending || !h->defer_open) return QH_OK;
  // (a chunk in flight is the one that filled the buckets last: its counters are still being gathered in d_meta)
  return launch_flush(h, h->pending);
}

static int write_all(int fd, const void* p, size_t n, off_t off) {
  const char* c = (const char*)p;
  while (n) {
    ssize_t w = pwrite(fd, c, n, off);
    if (w < 0) { if (errno == EINTR) continue; return -1; }
    c += w; n -= (size_t)w; off += w;
  }
  return 0;
}

// A large piece goes into the file through a shared mapping, copied by a few threads at once: buffered pwrite()s to
// one file serialise on its inode lock (8 threads: no gain measured), page faults on a mapping do not. Eight threads
// are the measured optimum on the GPU box (4.8 GB database: 0.40 s; 4 threads 1.13 s, 12 threads 1.93 s, 16 threads
// 0.60 s -- the faults contend for the file's page-cache lock); QH_IO_THREADS overrides.
static void copy_parallel(char* dst, const char* src, size_t n) {
  const size_t min_slice = (size_t)2 << 20;
  unsigned nt = std::thread::hardware_concurrency();
  if (nt > 8) nt = 8;
  if (const char* e = getenv("QH_IO_THREADS")) { const int v = atoi(e); if (v > 0) nt = (unsigned)v; }
  if (nt < 2 || n < 2 * min_slice) { memcpy(dst, src, n); return; }
  if (n / nt < min_slice) nt = (unsigned)(n / min_slice);
  const size_t slice = ((n + nt - 1) / nt + 4095) & ~(size_t)4095;
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t) {
    const size_t from = (size_t)t * slice, to = from + slice < n ? from + slice : n;
    if (from >= to) break;
    th.emplace_back([=]() { memcpy(dst + from, src + from, to - from); });
  }
  for (auto& x : th) x.join();
}

extern "C" int qh_build_write(qh_build* h, const char* path, int argc, const char* const* argv) {
  if (!h || !path) return qh_fail(QH_EBADARG, "null argument");
  int rc = qh_build_sync(h);
  if (rc) return rc;
  if (h->g.shard_bits) return qh_fail(QH_EBADARG, "qh_build_write on a sharded handle: use qh_build_write_shard");
  return qh_build_write_shard(h, path, argc, argv, 1);
}

// Every shard writes its slice of the keys image; values are packed per shard into
// the global atomic_bits_array image (word-aligned when slots % vpw == 0, else the
// shard owner ORs its fields into shared boundary words -- single writer at a time).
extern "C" int qh_build_write_shard(qh_build* h, const char* path, int argc, const char* const* argv, int write_header) {
  if (h) { int mrc = materialize(h); if (mrc) return mrc; }
  if (!h || !path) return qh_fail(QH_EBADARG, "null argument");
  int rc = qh_build_sync(h);
  if (rc) return rc;
  QH_HIP(hipSetDevice(h->device));
  qh_file_header fh;
  qh_header_from_geom(h->g, fh);
  qh_header_fill_standard(fh, argc, argv);
  std::string hdr;
  qh_header_serialize(fh, hdr);
  // read-write: the boundary words of the values image are completed by read-OR-write (see below)
  int fd = open(path, write_header ? (O_RDWR | O_CREAT | O_TRUNC) : O_RDWR, 0644);
  if (fd < 0) return qh_fail(QH_EIO, "Failed to open output file '%s'.", path);  // create_database.cc:113-115
  bool ok = true;
  if (write_header) {
    ok = write_all(fd, hdr.data(), hdr.size(), 0) == 0;
    // extend to the final length so that shards can pwrite in any order
    if (ok) ok = ftruncate(fd, (off_t)(fh.offset + fh.key_bytes + fh.value_bytes)) == 0;
  } else {
    // offset depends on the header text (hostname, cmdline...): read it back from the file
    qh_file_header on_disk;
    std::string err;
    if (!qh_header_read_file(path, on_disk, err)) { close(fd); return qh_fail(QH_EIO, "%s", err.c_str()); }
    fh.offset = on_disk.offset;
  }
  const uint64_t slots = 1ull << h->g.lsize;
  const uint64_t first = (uint64_t)h->g.shard_id * slots;
  const int vbits = h->g.bits + 1;
  const uint64_t vpw = 64 / vbits;
  const uint64_t w0 = first / vpw, w1 = (first + slots + vpw - 1) / vpw;  // value words touched by this shard
  // values image packed on the GPU while the keys image is on its way out
  uint64_t* d_vimg = nullptr;
  if (hipMalloc((void**)&d_vimg, (w1 - w0) * 8) != hipSuccess) {
    (void)hipGetLastError();
    close(fd);
    return qh_fail(QH_ENOMEM, "cannot allocate the %llu-byte values image", (unsigned long long)((w1 - w0) * 8));
  }
  qh_launch_vals_image(h->table, h->g, first, slots, w0, w1 - w0, d_vimg, h->stream);
  // keys image: two pinned staging pieces, the copy of piece i+1 overlaps the pwrite of piece i
  const size_t piece = slots < ((size_t)1 << 22) ? (size_t)slots : ((size_t)1 << 22);  // slots per piece (32 MiB)
  uint64_t* stage[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
  hipError_t e = hipSuccess;
  for (int i = 0; i < 2 && e == hipSuccess; ++i) {
    e = hipHostMalloc((void**)&stage[i], piece * 8, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
  }
  auto issue = [&](uint64_t at, int b) -> hipError_t {
    const uint64_t n = slots - at < piece ? slots - at : piece;
    hipError_t r = hipMemcpyAsync(stage[b], h->table + at, n * 8, hipMemcpyDeviceToHost, h->stream);
    return r == hipSuccess ? hipEventRecord(ev[b], h->stream) : r;
  };
  // this shard's slice of the keys image, mapped (page-aligned start); plain pwrite when the mapping fails
  const off_t key_at = (off_t)(fh.offset + first * 8);
  const off_t map_off = key_at & ~(off_t)4095;
  const size_t map_len = (size_t)(key_at - map_off) + (size_t)slots * 8;
  // (the blocks are reserved first: a full disk must be an error return here, not a SIGBUS in a copy thread)
  char* map = nullptr;
  if (posix_fallocate(fd, key_at, (off_t)slots * 8) == 0) {
    map = static_cast<char*>(mmap(nullptr, map_len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, map_off));
    if (map == MAP_FAILED) map = nullptr;
  }
  if (e == hipSuccess) e = issue(0, 0);
  int b = 0;
  for (uint64_t at = 0; ok && e == hipSuccess && at < slots; at += piece, b ^= 1) {
    const uint64_t n = slots - at < piece ? slots - at : piece;
    if (at + piece < slots) e = issue(at + piece, b ^ 1);
    if (e == hipSuccess) e = hipEventSynchronize(ev[b]);
    if (e == hipSuccess) {
      const off_t at_file = (off_t)(fh.offset + (first + at) * 8);
      if (map) copy_parallel(map + (at_file - map_off), (const char*)stage[b], n * 8);
      else ok = write_all(fd, stage[b], n * 8, at_file) == 0;
    }
  }
  if (map && munmap(map, map_len) != 0) ok = false;
  std::vector<uint64_t> vimg(w1 - w0, 0);
  if (e == hipSuccess) e = hipMemcpyAsync(vimg.data(), d_vimg, (w1 - w0) * 8, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  for (int i = 0; i < 2; ++i) {
    if (stage[i]) { hipError_t x = hipHostFree(stage[i]); (void)x; }
    if (ev[i]) { hipError_t x = hipEventDestroy(ev[i]); (void)x; }
  }
  { hipError_t x = hipFree(d_vimg); (void)x; }
  if (e != hipSuccess) { close(fd); return qh_fail(QH_EHIP, "reading the table back failed: %s", hipGetErrorString(e)); }
  if (ok) {
    const off_t voff = (off_t)(fh.offset + fh.key_bytes + w0 * 8);
    if (h->g.shard_bits && (first % vpw || (first + slots) % vpw)) {
      // boundary words are shared with the neighbouring shards: OR into what is there
      uint64_t edge;
      if (pread(fd, &edge, 8, voff) == 8) vimg.front() |= edge;
      else ok = false;
      if (vimg.size() > 1) {
        if (pread(fd, &edge, 8, voff + (off_t)(vimg.size() - 1) * 8) == 8) vimg.back() |= edge;
        else ok = false;
      }
    }
    // large values images go through a mapping too (edge words included: they were completed above)
    const size_t vbytes = vimg.size() * 8;
    bool done = false;
    if (ok && vbytes >= ((size_t)64 << 20) && posix_fallocate(fd, voff, (off_t)vbytes) == 0) {
      const off_t vmap_off = voff & ~(off_t)4095;
      const size_t vmap_len = (size_t)(voff - vmap_off) + vbytes;
      char* vmap = static_cast<char*>(mmap(nullptr, vmap_len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, vmap_off));
      if (vmap != MAP_FAILED) {
        copy_parallel(vmap + (voff - vmap_off), (const char*)vimg.data(), vbytes);
        if (munmap(vmap, vmap_len) != 0) ok = false;
        done = true;
      }
    }
    if (ok && !done) ok = write_all(fd, vimg.data(), vbytes, voff) == 0;
  }
  if (close(fd) != 0) ok = false;
  if (!ok) return qh_fail(QH_EIO, "Failed to write output file '%s': %s", path, strerror(errno));
  return QH_OK;
}

extern "C" int qh_build_dump(qh_build* h, uint64_t* mers, uint64_t* vals, uint64_t cap, uint64_t* n) {
  if (h) { int mrc = materialize(h); if (mrc) return mrc; }
  if (!h || !n) return qh_fail(QH_EBADARG, "null argument");
  int rc = qh_build_sync(h);
  if (rc) return rc;
  return qh_table_dump(h->device, h->stream, h->table, h->g, h->d_meta, h->h_meta, mers, vals, cap, n);
}

// The database in the recalled Jellyfish layout (qh_jflayout.h; parity unpinned): the table's entries are read
// back and the bit-packed image is built on the host -- a serialisation switch, not a hot path.
extern "C" int qh_build_write_jf(qh_build* h, const char* path, int argc, const char* const* argv) {
  if (!h || !path) return qh_fail(QH_EBADARG, "null argument");
  if (h->g.shard_bits) return qh_fail(QH_EBADARG, "qh_build_write_jf on a sharded handle");
  qh_build_stats st;
  int rc = qh_build_finish(h, &st);
  if (rc) return rc;
  std::vector<uint64_t> mers(st.distinct ? st.distinct : 1), vals(st.distinct ? st.distinct : 1);
  uint64_t n = 0;
  if ((rc = qh_build_dump(h, mers.data(), vals.data(), st.distinct, &n))) return rc;
  if (n != st.distinct) return qh_fail(QH_EHIP, "the table holds %llu entries, %llu were counted", (unsigned long long)n, (unsigned long long)st.distinct);
  return qh_jf_write_file(path, h->g.k, h->g.bits, 1ull << h->g.lsize, h->reprobe_limit, mers.data(), vals.data(), n, argc, argv);
}

extern "C" int qh_build_as_query(qh_build* h, qh_query** out) {
  if (h) { int mrc = materialize(h); if (mrc) return mrc; }
  if (!h || !out) return qh_fail(QH_EBADARG, "null argument");
  int rc = qh_build_sync(h);
  if (rc) return rc;
  if (h->g.shard_bits) return qh_fail(QH_EBADARG, "a sharded build handle cannot serve lookups on its own");
  return qh_query_from_table(h->device, h->table, h->g, out);
}

extern "C" int qh_detect_min_qual_char(const char* buf, size_t len, int* min_q_char) {
  if (!buf || !min_q_char) return qh_fail(QH_EBADARG, "null argument");
  int minq = 256;
  size_t p = 0, line = 0;
  auto next_line = [&](size_t* b, size_t* e) -> bool {  // perl's <$io>: false at end of input
    if (p >= len) return false;
    const void* nl = memchr(buf + p, '\n', len - p);
    *b = p;
    *e = nl ? (size_t)((const char*)nl - buf) : len;
    p = nl ? *e + 1 : len;
    ++line;
    return true;
  };
  size_t b, e;
  while (next_line(&b, &e)) {            // header
    if (line > 4000) break;
    next_line(&b, &e);                   // sequence
    next_line(&b, &e);                   // quality header
    if (!next_line(&b, &e)) return qh_fail(QH_EFORMAT, "Invalid fastq format");
    for (size_t i = b; i < e; ++i) {
      const int c = (unsigned char)buf[i];
      if (c < minq) minq = c;
    }
  }
  if (minq == 35 || minq == 66) minq -= 2;  // special Illumina case where quality 0 and 1 do not appear
  *min_q_char = minq;
  if (minq != 33 && minq != 59 && minq != 64)
    return qh_fail(QH_EFORMAT, "Found an unusual minimum quality char of %d (%c). Stopping now. Use option -m to override",
                   minq, minq >= 32 && minq < 127 ? minq : '?');
  return QH_OK;
}

extern "C" int qh_mer_hash(int k, const uint64_t* mers, size_t n, uint64_t* hashes) {
  if (k < 1 || k > 31) return qh_fail(QH_EBADARG, "Mer length must be between 1 and 31 (got %d)", k);
  if (n && (!mers || !hashes)) return qh_fail(QH_EBADARG, "null buffer");
  const uint64_t kmask = qh_low_mask(2 * k);
  for (size_t i = 0; i < n; ++i) hashes[i] = qh_hash_key(mers[i] & kmask, k);
  return QH_OK;
}

extern "C" int qh_host_alloc(void** out, size_t bytes) {
  if (!out || !bytes) return qh_fail(QH_EBADARG, "null argument");
  *out = nullptr;
  hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
  if (e != hipSuccess) { (void)hipGetLastError(); return qh_fail(QH_ENOMEM, "cannot pin %zu bytes of host memory: %s", bytes, hipGetErrorString(e)); }
  return QH_OK;
}

extern "C" void qh_host_free(void* p) {
  if (p) { hipError_t e = hipHostFree(p); (void)e; }
}

// number of leading bytes of buf that form whole groups of 4 lines (the chunk cut
// used by the CLI's fast path); a last line without '\n' counts only when `final_chunk`
// The same cut from the chunk's last lines alone. In strict 4-line FASTQ a line that starts with '@' and whose
// second successor starts with '+' is a header and nothing else (a quality line may start with '@', but two lines
// after it comes a sequence, which cannot start with '+'), so the last such line among the final dozen fixes the
// phase of the records and the cut follows without counting the newlines of the whole chunk (40 M memchr calls on a
// 2 GB file: 0.25 s, most of what the tools spent outside file I/O). Returns false when those lines do not settle it
// (very long or too few lines, other formats): the caller then counts from the start. A chunk that is not strict
// 4-line FASTQ is rejected by the GPU indexer whatever cut was chosen.
static bool prefix_from_tail(const char* buf, size_t len, int final_chunk, size_t* out) {
  struct line { size_t s, e; bool term; };  // [s, e) without its newline; term: the newline is in the chunk
  enum { MAXL = 12 };
  line rev[MAXL];
  int n = 0;
  if (len < 2) return false;
  size_t e = len;
  bool term = false;
  if (buf[len - 1] == '\n') { e = len - 1; term = true; }
  bool reached_start = false;
  while (n < MAXL) {
    const void* q = e ? memrchr(buf, '\n', e) : nullptr;
    const size_t s = q ? (size_t)((const char*)q - buf) + 1 : 0;
    rev[n++] = line{s, e, term};
    if (!q) { reached_start = true; break; }
    e = s - 1;
    term = true;
  }
  auto at = [&](int i) -> const line& { return rev[n - 1 - i]; };  // ascending: at(0) is the earliest line seen
  auto starts = [&](int i, char c) { return at(i).e > at(i).s && buf[at(i).s] == c; };
  int h = -1;
  for (int i = n - 3; i >= 0; --i)
    if (starts(i, '@') && starts(i + 2, '+')) { h = i; break; }
  if (h < 0) return false;
  // a chunk seen from its very first line must agree with the count from the start: its first line is a header
  if (reached_start && (h & 3) != 0) return false;
  int g = h;
  while (g + 3 < n && (at(g + 3).term || (final_chunk && g + 3 == n - 1))) g += 4;
  *out = g < n ? at(g).s : len;
  return true;
}

extern "C" size_t qh_fastq_4line_prefix(const char* buf, size_t len, int final_chunk) {
  size_t cut = 0;
  if (!getenv("QH_PREFIX_BY_COUNTING") && prefix_from_tail(buf, len, final_chunk, &cut)) return cut;
  size_t lines = 0, pos = 0, last_group_end = 0;
  while (pos < len) {
    const void* e = memchr(buf + pos, '\n', len - pos);
    if (!e) {
      if (final_chunk && ((lines + 1) & 3) == 0) last_group_end = len;
      break;
    }
    pos = (size_t)((const char*)e - buf) + 1;
    if ((++lines & 3) == 0) last_group_end = pos;
  }
  return last_group_end;
}
