// qh_query.hip -- read side of the C ABI: database_query (src/mer_database.hpp:251-362)
// as batched GPU lookups over the table resident in HBM, plus the whole-table
// reductions its callers run (src/error_correct_reads.cc:650-659,
// src/histo_mer_database.cc:13-21) and the roofline microbenchmark.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include <errno.h>
#include <sys/stat.h>
#include <new>
#include <thread>
#include <vector>

#include "qh_internal.h"
#include "qh_dbfile.h"
#include "qh_jflayout.h"

int qh_ensure_device(void** p, size_t* cap, size_t need, const char* what) {
  if (*cap >= need && *p) return QH_OK;
  if (*p) { hipError_t e = hipFree(*p); (void)e; *p = nullptr; *cap = 0; }
  const size_t want = need + need / 8 + 256;
  hipError_t e = hipMalloc(p, want);
  if (e != hipSuccess) { *p = nullptr; return qh_fail(QH_ENOMEM, "hipMalloc of %zu bytes for %s failed: %s", want, what, hipGetErrorString(e)); }
  *cap = want;
  return QH_OK;
}

static int query_init_common(qh_query* q) {
  QH_HIP(hipSetDevice(q->device));
  QH_HIP(hipStreamCreateWithFlags(&q->stream, hipStreamNonBlocking));
  QH_HIP(hipMalloc((void**)&q->d_meta, sizeof(qh_meta)));
  QH_HIP(hipHostMalloc((void**)&q->h_meta, sizeof(qh_meta), hipHostMallocDefault));
  return QH_OK;
}

int qh_query_from_table(int device, uint64_t* table, const qh_geom& g, qh_query** out) {
  qh_query* q = new qh_query();
  q->device = device; q->g = g; q->table = table; q->owns_table = false;
  int rc = query_init_common(q);
  if (rc) { qh_query_close(q); return rc; }
  *out = q;
  return QH_OK;
}

// A file in the jellyfish layout (qh_jflayout.h, parity unpinned): its entries are decoded on the host and
// re-inserted, values carried, into a native table of the same size in HBM (doubled while a key finds no place).
static int open_jf_layout(qh_query** out, const char* path, int device) {
  std::string err;
  std::vector<uint64_t> mers, vals;
  int k = 0, bits = 0;
  uint64_t size = 0;
  if (!qh_jf_read_pairs(path, &k, &bits, &size, mers, vals, err)) return qh_fail(QH_EFORMAT, "%s for file '%s'", err.c_str(), path);
  qh_query* q = new qh_query();
  q->device = device;
  int rc = query_init_common(q);
  if (rc) { qh_query_close(q); return rc; }
  const uint64_t n = mers.size();
  uint64_t *d_m = nullptr, *d_v = nullptr;
  auto cleanup = [&]() { hipError_t x = hipFree(d_m); x = hipFree(d_v); (void)x; };
  if (n) {
    hipError_t e = hipMalloc((void**)&d_m, n * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&d_v, n * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(d_m, mers.data(), n * 8, hipMemcpyHostToDevice, q->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_v, vals.data(), n * 8, hipMemcpyHostToDevice, q->stream);
    if (e != hipSuccess) { (void)hipGetLastError(); cleanup(); qh_query_close(q); return qh_fail(QH_ENOMEM, "cannot stage %llu entries on the device: %s", (unsigned long long)n, hipGetErrorString(e)); }
  }
  for (uint64_t slots = size;; slots <<= 1) {
    qh_geom g;
    if (slots > (1ull << 44) || !qh_make_geom(k, bits, slots, 0, 126, 0, 0, -1, g, err)) { cleanup(); qh_query_close(q); return qh_fail(QH_EFULL, "Hash is full"); }
    const size_t bytes = (size_t)8 << g.lsize;
    hipError_t e = hipMalloc((void**)&q->table, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(q->table, 0, bytes, q->stream);
    if (e == hipSuccess) e = hipMemsetAsync(q->d_meta, 0, sizeof(qh_meta), q->stream);
    if (e == hipSuccess && n) { qh_launch_insert_valued(d_m, d_v, n, q->table, g, q->d_meta, q->stream); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(q->h_meta, q->d_meta, sizeof(qh_meta), hipMemcpyDeviceToHost, q->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(q->stream);
    if (e != hipSuccess) { (void)hipGetLastError(); cleanup(); qh_query_close(q); return qh_fail(QH_EHIP, "building the table from '%s' failed: %s", path, hipGetErrorString(e)); }
    q->g = g;
    if (!q->h_meta->rehash_fail) break;
    e = hipFree(q->table); (void)e;
    q->table = nullptr;
  }
  cleanup();
  *out = q;
  return QH_OK;
}

extern "C" int qh_query_open(qh_query** out, const char* path, int device) {
  if (!out || !path) return qh_fail(QH_EBADARG, "null argument");
  *out = nullptr;
  qh_file_header fh;
  std::string err;
  if (!qh_header_read_file(path, fh, err)) return qh_fail(QH_EIO, "%s", err.c_str());
  if (fh.format == "binary/quorum_db" && fh.layout == QH_LAYOUT_JF) {
    try {
      return open_jf_layout(out, path, device);
    } catch (const std::bad_alloc&) {  // no exception crosses the C ABI (a header may ask for more than the host has)
      return qh_fail(QH_ENOMEM, "out of host memory reading '%s'", path);
    }
  }
  qh_geom g;
  if (!qh_geom_from_header(fh, g, err)) return qh_fail(QH_EFORMAT, "%s for file '%s'", err.c_str(), path);
  int fd = open(path, O_RDONLY);
  if (fd < 0) return qh_fail(QH_EIO, "Can't open '%s' for reading", path);
  struct stat sb;
  if (fstat(fd, &sb) != 0 || (uint64_t)sb.st_size < fh.offset + fh.key_bytes + fh.value_bytes) {
    close(fd);
    return qh_fail(QH_EIO, "file '%s' is shorter than its header says", path);
  }
  qh_query* q = new qh_query();
  q->device = device; q->g = g;
  int rc = query_init_common(q);
  if (rc) { close(fd); qh_query_close(q); return rc; }
  hipError_t e = hipMalloc((void**)&q->table, fh.key_bytes);
  if (e != hipSuccess) { close(fd); qh_query_close(q); return qh_fail(QH_ENOMEM, "cannot allocate %llu bytes of HBM for the table: %s", (unsigned long long)fh.key_bytes, hipGetErrorString(e)); }
  // stream the keys image into HBM through two pinned staging pieces: piece i + 1 is read from the file (by a few
  // threads at once: the page-cache copy is what takes the time) while piece i travels to the device
  const size_t piece = (size_t)64 << 20;
  void* stage[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
  for (int i = 0; i < 2 && e == hipSuccess; ++i) {
    e = hipHostMalloc(&stage[i], piece, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
  }
  if (e != hipSuccess) {
    for (int i = 0; i < 2; ++i) { if (stage[i]) (void)hipHostFree(stage[i]); if (ev[i]) (void)hipEventDestroy(ev[i]); }
    close(fd); qh_query_close(q);
    return qh_fail(QH_ENOMEM, "cannot allocate the staging buffers");
  }
  auto read_piece = [&](char* dst, size_t n, uint64_t file_at) -> bool {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 8) nt = 8;
    if (nt < 1 || n < ((size_t)16 << 20)) nt = 1;
    const size_t slice = ((n + nt - 1) / nt + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    std::vector<char> bad(nt, 0);
    for (unsigned t = 0; t < nt; ++t) {
      const size_t from = (size_t)t * slice, to = from + slice < n ? from + slice : n;
      if (from >= to) break;
      th.emplace_back([=, &bad]() {
        size_t got = from;
        while (got < to) {
          ssize_t r = pread(fd, dst + got, to - got, (off_t)(file_at + got));
          if (r < 0 && errno == EINTR) continue;
          if (r <= 0) { bad[t] = 1; return; }
          got += (size_t)r;
        }
      });
    }
    for (auto& x : th) x.join();
    for (char c : bad) if (c) return false;
    return true;
  };
  rc = QH_OK;
  bool used[2] = {false, false};
  int b = 0;
  // `bytes` of the file from `file_off` on -> device memory at dst
  auto stream_in = [&](char* dst, uint64_t file_off, uint64_t bytes) {
    for (uint64_t at = 0; at < bytes && rc == QH_OK; at += piece, b ^= 1) {
      const size_t n = bytes - at < piece ? (size_t)(bytes - at) : piece;
      if (used[b]) {  // the copy that last read this piece must be done before the file overwrites it
        e = hipEventSynchronize(ev[b]);
        if (e != hipSuccess) { rc = qh_fail(QH_EHIP, "copying the table to the device failed: %s", hipGetErrorString(e)); break; }
      }
      if (!read_piece((char*)stage[b], n, file_off + at)) { rc = qh_fail(QH_EIO, "Failed to read in file '%s'", path); break; }
      e = hipMemcpyAsync(dst + at, stage[b], n, hipMemcpyHostToDevice, q->stream);
      if (e == hipSuccess) e = hipEventRecord(ev[b], q->stream);
      used[b] = true;
      if (e != hipSuccess) rc = qh_fail(QH_EHIP, "copying the table to the device failed: %s", hipGetErrorString(e));
    }
  };
  stream_in((char*)q->table, fh.offset, fh.key_bytes);
  uint64_t* d_vimg = nullptr;
  if (rc == QH_OK && g.wide) {
    // wide form: the slots hold keys only; the values image (atomic_bits_array layout) supplies the side array
    const int vb = g.bits + 1;
    e = hipMalloc((void**)&q->g.wvals, fh.key_bytes);
    if (e == hipSuccess && vb <= 32) e = hipMalloc((void**)&d_vimg, fh.value_bytes);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      rc = qh_fail(QH_ENOMEM, "cannot allocate %llu bytes of HBM for the values: %s", (unsigned long long)fh.key_bytes, hipGetErrorString(e));
    } else if (vb > 32) {
      stream_in((char*)q->g.wvals, fh.offset + fh.key_bytes, fh.value_bytes);  // one field per word: the image is the array
    } else {
      stream_in((char*)d_vimg, fh.offset + fh.key_bytes, fh.value_bytes);
      if (rc == QH_OK) {
        qh_launch_vals_unpack(d_vimg, vb, fh.size, q->g.wvals, q->stream);
        e = hipGetLastError();
        if (e != hipSuccess) rc = qh_fail(QH_EHIP, "unpacking the values image failed: %s", hipGetErrorString(e));
      }
    }
  }
  e = hipStreamSynchronize(q->stream);
  if (e != hipSuccess && rc == QH_OK) rc = qh_fail(QH_EHIP, "copying the table to the device failed: %s", hipGetErrorString(e));
  for (int i = 0; i < 2; ++i) { (void)hipHostFree(stage[i]); (void)hipEventDestroy(ev[i]); }
  if (d_vimg) (void)hipFree(d_vimg);
  close(fd);
  if (rc) { qh_query_close(q); return rc; }
  *out = q;
  return QH_OK;
}

extern "C" int qh_db_header_info(const char* path, qh_db_info* info) {
  if (!path || !info) return qh_fail(QH_EBADARG, "null argument");
  qh_file_header fh;
  std::string err;
  if (!qh_header_read_file(path, fh, err)) return qh_fail(QH_EIO, "%s", err.c_str());
  memset(info, 0, sizeof *info);
  info->size = fh.size; info->key_len = fh.key_len; info->val_len = fh.val_len;
  info->max_reprobe = fh.max_reprobe; info->bits = fh.bits;
  info->key_bytes = fh.key_bytes; info->value_bytes = fh.value_bytes; info->offset = fh.offset;
  info->shard_bits = fh.shard_bits; info->window_bits = fh.window_bits;
  snprintf(info->format, sizeof info->format, "%s", fh.format.c_str());
  if (fh.format != "binary/quorum_db") return qh_fail(QH_EFORMAT, "Wrong type '%s' for file '%s'", fh.format.c_str(), path);
  return QH_OK;
}

extern "C" int qh_db_header_parse(const void* data, size_t len, qh_db_info* info) {
  if (!data || !info) return qh_fail(QH_EBADARG, "null argument");
  qh_file_header fh;
  std::string err;
  if (!qh_header_parse(static_cast<const char*>(data), len, fh, err)) return qh_fail(QH_EFORMAT, "Can't parse header: %s", err.c_str());
  memset(info, 0, sizeof *info);
  info->size = fh.size; info->key_len = fh.key_len; info->val_len = fh.val_len;
  info->max_reprobe = fh.max_reprobe; info->bits = fh.bits;
  info->key_bytes = fh.key_bytes; info->value_bytes = fh.value_bytes; info->offset = fh.offset;
  info->shard_bits = fh.shard_bits; info->window_bits = fh.window_bits;
  snprintf(info->format, sizeof info->format, "%s", fh.format.c_str());
  if (fh.format != "binary/quorum_db") return qh_fail(QH_EFORMAT, "Wrong type '%s'", fh.format.c_str());
  return QH_OK;
}

extern "C" void qh_query_close(qh_query* q) {
  if (!q) return;
  hipError_t e = hipSetDevice(q->device); (void)e;
  if (q->stream) { e = hipStreamSynchronize(q->stream); (void)e; }
  void* ptrs[] = {q->owns_table ? (void*)q->table : nullptr, q->owns_table ? (void*)q->g.wvals : nullptr, q->d_meta, q->d_out};
  for (void* p : ptrs) if (p) { e = hipFree(p); (void)e; }
  for (qh_query::lane* l : q->lanes) {
    if (l->stream) { e = hipStreamSynchronize(l->stream); (void)e; }
    void* lp[] = {l->d_in, l->d_in2, l->d_out};
    for (void* p : lp) if (p) { e = hipFree(p); (void)e; }
    if (l->stream) { e = hipStreamDestroy(l->stream); (void)e; }
    delete l;
  }
  if (q->h_meta) { e = hipHostFree(q->h_meta); (void)e; }
  if (q->stream) { e = hipStreamDestroy(q->stream); (void)e; }
  delete q;
}

extern "C" int qh_query_info(const qh_query* q, int* k, int* bits, uint64_t* size) {
  if (!q) return qh_fail(QH_EBADARG, "null handle");
  if (k) *k = q->g.k;
  if (bits) *bits = q->g.bits;
  if (size) *size = qh_table_slots(q->g);
  return QH_OK;
}

extern "C" int qh_query_batch_device(qh_query* q, const uint64_t* dev_mers, size_t n, uint64_t* dev_out, void* hip_stream) {
  if (!q) return qh_fail(QH_EBADARG, "null handle");
  if (n == 0) return QH_OK;
  if (!dev_mers || !dev_out) return qh_fail(QH_EBADARG, "null buffer");
  QH_HIP(hipSetDevice(q->device));
  qh_launch_lookup(q->table, q->g, dev_mers, n, dev_out, 0, hip_stream ? (hipStream_t)hip_stream : q->stream);
  QH_HIP(hipGetLastError());
  return QH_OK;
}

// a lane of the handle for the duration of one call (qh_internal.h)
namespace {
class lane_guard {
  qh_query* q_;
  qh_query::lane* l_ = nullptr;
public:
  explicit lane_guard(qh_query* q) : q_(q) {
    std::unique_lock<std::mutex> lk(q->lane_mu);
    for (;;) {
      for (qh_query::lane* l : q->lanes) if (!l->busy) { l_ = l; break; }
      if (l_) break;
      if (q->lanes.size() < QH_QUERY_MAX_LANES) {
        qh_query::lane* l = new qh_query::lane();
        if (hipSetDevice(q->device) != hipSuccess || hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) != hipSuccess) {
          (void)hipGetLastError();
          delete l;
          if (q->lanes.empty()) return;  // no lane at all: the caller reports the failure
        } else {
          q->lanes.push_back(l);
          l_ = l;
          break;
        }
      }
      q->lane_cv.wait(lk);
    }
    l_->busy = true;
  }
  ~lane_guard() {
    if (!l_) return;
    { std::lock_guard<std::mutex> lk(q_->lane_mu); l_->busy = false; }
    q_->lane_cv.notify_one();
  }
  qh_query::lane* get() const { return l_; }
};
}  // namespace

static int lookup_host(qh_query* q, const uint64_t* mers, size_t n, uint64_t* out, int get_val) {
  if (!q) return qh_fail(QH_EBADARG, "null handle");
  if (n == 0) return QH_OK;
  if (!mers || !out) return qh_fail(QH_EBADARG, "null buffer");
  lane_guard lg(q);
  qh_query::lane* l = lg.get();
  if (!l) return qh_fail(QH_EHIP, "cannot create a stream for the lookup");
  QH_HIP(hipSetDevice(q->device));
  int rc;
  if ((rc = qh_ensure_device(&l->d_in, &l->in_cap, n * 8, "query keys"))) return rc;
  if ((rc = qh_ensure_device(&l->d_out, &l->out_cap, n * 8, "query results"))) return rc;
  QH_HIP(hipMemcpyAsync(l->d_in, mers, n * 8, hipMemcpyHostToDevice, l->stream));
  qh_launch_lookup(q->table, q->g, (const uint64_t*)l->d_in, n, (uint64_t*)l->d_out, get_val, l->stream);
  QH_HIP(hipGetLastError());
  QH_HIP(hipMemcpyAsync(out, l->d_out, n * 8, hipMemcpyDeviceToHost, l->stream));
  QH_HIP(hipStreamSynchronize(l->stream));
  return QH_OK;
}

extern "C" int qh_query_batch(qh_query* q, const uint64_t* mers, size_t n, uint64_t* out) {
  return lookup_host(q, mers, n, out, 0);
}

extern "C" int qh_query_get_val_batch(qh_query* q, const uint64_t* mers, size_t n, uint64_t* out) {
  return lookup_host(q, mers, n, out, 1);
}

extern "C" int qh_query_best_alternatives_batch(qh_query* q, const uint64_t* fmers, const uint8_t* backward, size_t n,
                                                uint64_t* counts, int32_t* ucode, int32_t* level, int32_t* nkept) {
  if (!q) return qh_fail(QH_EBADARG, "null handle");
  if (n == 0) return QH_OK;
  if (!fmers || !backward || !counts || !ucode || !level || !nkept) return qh_fail(QH_EBADARG, "null buffer");
  lane_guard lg(q);
  qh_query::lane* l = lg.get();
  if (!l) return qh_fail(QH_EHIP, "cannot create a stream for the lookup");
  QH_HIP(hipSetDevice(q->device));
  int rc;
  // d_in: fmers; d_in2: backward flags; d_out: counts[n][4] then ucode, level, nkept
  if ((rc = qh_ensure_device(&l->d_in, &l->in_cap, n * 8, "query keys"))) return rc;
  if ((rc = qh_ensure_device(&l->d_in2, &l->in2_cap, n, "direction flags"))) return rc;
  if ((rc = qh_ensure_device(&l->d_out, &l->out_cap, n * (32 + 12), "query results"))) return rc;
  uint64_t* d_counts = (uint64_t*)l->d_out;
  int32_t* d_ucode = (int32_t*)(d_counts + n * 4);
  int32_t* d_level = d_ucode + n;
  int32_t* d_nkept = d_level + n;
  QH_HIP(hipMemcpyAsync(l->d_in, fmers, n * 8, hipMemcpyHostToDevice, l->stream));
  QH_HIP(hipMemcpyAsync(l->d_in2, backward, n, hipMemcpyHostToDevice, l->stream));
  qh_launch_best_alternatives(q->table, q->g, (const uint64_t*)l->d_in, (const uint8_t*)l->d_in2, n, d_counts, d_ucode,
                              d_level, d_nkept, l->stream);
  QH_HIP(hipGetLastError());
  QH_HIP(hipMemcpyAsync(counts, d_counts, n * 32, hipMemcpyDeviceToHost, l->stream));
  QH_HIP(hipMemcpyAsync(ucode, d_ucode, n * 4, hipMemcpyDeviceToHost, l->stream));
  QH_HIP(hipMemcpyAsync(level, d_level, n * 4, hipMemcpyDeviceToHost, l->stream));
  QH_HIP(hipMemcpyAsync(nkept, d_nkept, n * 4, hipMemcpyDeviceToHost, l->stream));
  QH_HIP(hipStreamSynchronize(l->stream));
  return QH_OK;
}

extern "C" int qh_query_stats(qh_query* q, uint64_t* distinct, uint64_t* total) {
  if (!q || !distinct || !total) return qh_fail(QH_EBADARG, "null argument");
  std::lock_guard<std::mutex> lock(q->mu);
  QH_HIP(hipSetDevice(q->device));
  QH_HIP(hipMemsetAsync(q->d_meta, 0, sizeof(qh_meta), q->stream));
  qh_launch_stats(q->table, q->g, q->d_meta, q->stream);
  QH_HIP(hipGetLastError());
  QH_HIP(hipMemcpyAsync(q->h_meta, q->d_meta, sizeof(qh_meta), hipMemcpyDeviceToHost, q->stream));
  QH_HIP(hipStreamSynchronize(q->stream));
  *distinct = q->h_meta->stat_distinct;
  *total = q->h_meta->stat_total;
  return QH_OK;
}

// poisson_term (error_correct_reads.cc:53-61)
extern "C" double qh_poisson_term(double lambda, unsigned i) {
  static const double facts[11] = {1, 1, 2, 6, 24, 120, 720, 5040, 40320, 362880, 3628800};
  static const double tau = 6.283185307179583;
  if (i < 11) return exp(-lambda) * pow(lambda, i) / facts[i];
  return exp(-lambda + i) * pow(lambda / i, i) / sqrt(tau * i);
}

// compute_poisson_cutoff__ (error_correct_reads.cc:650-668): the full-table reduction runs on
// the GPU (k_stats), the <= 998 Poisson terms on the host.
extern "C" int qh_query_poisson_cutoff(qh_query* q, double collision_prob, double poisson_threshold, unsigned* cutoff) {
  if (!q || !cutoff) return qh_fail(QH_EBADARG, "null argument");
  uint64_t distinct = 0, total = 0;
  int rc = qh_query_stats(q, &distinct, &total);
  if (rc) return rc;
  const double coverage = (double)total / (double)distinct;
  const double lambda = coverage * collision_prob;
  *cutoff = 0;
  for (unsigned x = 2; x < 1000; ++x)
    if (qh_poisson_term(lambda, x) < poisson_threshold) { *cutoff = x + 1; break; }
  return QH_OK;
}

extern "C" int qh_query_batch_contaminant_device(qh_query* q, qh_query* contaminant, const uint64_t* dev_mers, size_t n,
                                                 uint64_t* dev_out, uint8_t* dev_flags, void* hip_stream) {
  if (!q || !contaminant) return qh_fail(QH_EBADARG, "null handle");
  if (n == 0) return QH_OK;
  if (!dev_mers || !dev_flags) return qh_fail(QH_EBADARG, "null buffer");
  if (q->g.k != contaminant->g.k)  // error_correct_reads.cc:703-705
    return qh_fail(QH_EBADARG, "Contaminant mer length (%d) different than correction mer length (%d)", contaminant->g.k, q->g.k);
  if (q->device != contaminant->device) return qh_fail(QH_EBADARG, "database and contaminant set live on different devices");
  QH_HIP(hipSetDevice(q->device));
  qh_launch_lookup_contaminant(q->table, q->g, contaminant->table, contaminant->g, dev_mers, n, dev_out, dev_flags,
                               hip_stream ? (hipStream_t)hip_stream : q->stream);
  QH_HIP(hipGetLastError());
  return QH_OK;
}

extern "C" int qh_query_batch_contaminant(qh_query* q, qh_query* contaminant, const uint64_t* mers, size_t n, uint64_t* out,
                                          uint8_t* flags) {
  if (!q || !contaminant) return qh_fail(QH_EBADARG, "null handle");
  if (n == 0) return QH_OK;
  if (!mers || !flags) return qh_fail(QH_EBADARG, "null buffer");
  lane_guard lg(q);
  qh_query::lane* l = lg.get();
  if (!l) return qh_fail(QH_EHIP, "cannot create a stream for the lookup");
  QH_HIP(hipSetDevice(q->device));
  int rc;
  if ((rc = qh_ensure_device(&l->d_in, &l->in_cap, n * 8, "query keys"))) return rc;
  if ((rc = qh_ensure_device(&l->d_in2, &l->in2_cap, n, "contaminant flags"))) return rc;
  if ((rc = qh_ensure_device(&l->d_out, &l->out_cap, n * 8, "query results"))) return rc;
  QH_HIP(hipMemcpyAsync(l->d_in, mers, n * 8, hipMemcpyHostToDevice, l->stream));
  if ((rc = qh_query_batch_contaminant_device(q, contaminant, (const uint64_t*)l->d_in, n, out ? (uint64_t*)l->d_out : nullptr,
                                              (uint8_t*)l->d_in2, l->stream)))
    return rc;
  if (out) QH_HIP(hipMemcpyAsync(out, l->d_out, n * 8, hipMemcpyDeviceToHost, l->stream));
  QH_HIP(hipMemcpyAsync(flags, l->d_in2, n, hipMemcpyDeviceToHost, l->stream));
  QH_HIP(hipStreamSynchronize(l->stream));
  return QH_OK;
}

// contaminant_database::is_contaminant (error_correct_reads.cc:94) for a batch
extern "C" int qh_query_is_member_batch(qh_query* set, const uint64_t* mers, size_t n, uint8_t* flags) {
  return qh_query_batch_contaminant(set, set, mers, n, nullptr, flags);
}

extern "C" int qh_query_histo(qh_query* q, uint64_t* histo) {
  if (!q || !histo) return qh_fail(QH_EBADARG, "null argument");
  std::lock_guard<std::mutex> lock(q->mu);
  QH_HIP(hipSetDevice(q->device));
  const size_t bytes = 1001 * 2 * sizeof(uint64_t);
  int rc;
  if ((rc = qh_ensure_device(&q->d_out, &q->out_cap, bytes, "histogram"))) return rc;
  QH_HIP(hipMemsetAsync(q->d_out, 0, bytes, q->stream));
  qh_launch_histo(q->table, q->g, (unsigned long long*)q->d_out, q->stream);
  QH_HIP(hipGetLastError());
  QH_HIP(hipMemcpyAsync(histo, q->d_out, bytes, hipMemcpyDeviceToHost, q->stream));
  QH_HIP(hipStreamSynchronize(q->stream));
  return QH_OK;
}

int qh_table_dump(int device, hipStream_t stream, const uint64_t* table, const qh_geom& g, qh_meta* d_meta, qh_meta* h_meta,
                  uint64_t* mers, uint64_t* vals, uint64_t cap, uint64_t* n) {
  QH_HIP(hipSetDevice(device));
  if (cap && (!mers || !vals)) return qh_fail(QH_EBADARG, "null output");
  uint64_t* d_mers = nullptr;
  uint64_t* d_vals = nullptr;
  if (cap) {
    QH_HIP(hipMalloc((void**)&d_mers, cap * 8));
    hipError_t e = hipMalloc((void**)&d_vals, cap * 8);
    if (e != hipSuccess) { (void)hipFree(d_mers); return qh_fail(QH_ENOMEM, "cannot allocate the dump buffers"); }
  }
  hipError_t e = hipMemsetAsync(d_meta, 0, sizeof(qh_meta), stream);
  if (e == hipSuccess) {
    qh_launch_dump(table, g, d_mers, d_vals, cap, d_meta, stream);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(h_meta, d_meta, sizeof(qh_meta), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e == hipSuccess) {
    *n = h_meta->dump_count;
    const uint64_t m = *n < cap ? *n : cap;
    if (m) {
      e = hipMemcpy(mers, d_mers, m * 8, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(vals, d_vals, m * 8, hipMemcpyDeviceToHost);
    }
  }
  if (d_mers) (void)hipFree(d_mers);
  if (d_vals) (void)hipFree(d_vals);
  if (e != hipSuccess) return qh_fail(QH_EHIP, "table dump failed: %s", hipGetErrorString(e));
  return QH_OK;
}

extern "C" int qh_query_dump(qh_query* q, uint64_t* mers, uint64_t* vals, uint64_t cap, uint64_t* n) {
  if (!q || !n) return qh_fail(QH_EBADARG, "null argument");
  std::lock_guard<std::mutex> lock(q->mu);
  return qh_table_dump(q->device, q->stream, q->table, q->g, q->d_meta, q->h_meta, mers, vals, cap, n);
}

// --------------------------------------------------------- roofline microbenchmark

extern "C" int qh_bench_random_access(int device, int lsize, uint64_t n_updates, int mode, int iters, double* ms) {
  if (!ms || lsize < 4 || lsize > 36 || iters < 1 || mode < 0 || mode > 2) return qh_fail(QH_EBADARG, "bad argument");
  QH_HIP(hipSetDevice(device));
  uint64_t* table = nullptr;
  unsigned long long* sink = nullptr;
  const size_t bytes = (size_t)8 << lsize;
  QH_HIP(hipMalloc((void**)&table, bytes));
  hipError_t e = hipMalloc((void**)&sink, 8);
  hipStream_t s = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e == hipSuccess) e = hipMemsetAsync(table, 0, bytes, s);
  float total = 0;
  for (int it = -1; it < iters && e == hipSuccess; ++it) {  // it == -1: warm-up
    e = hipEventRecord(e0, s);
    qh_launch_random_access(table, lsize, n_updates, mode, 0x1234u + (uint64_t)(it + 1) * 7919u, sink, s);
    if (e == hipSuccess) e = hipEventRecord(e1, s);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float t = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, e0, e1);
    if (it >= 0) total += t;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (s) (void)hipStreamDestroy(s);
  if (sink) (void)hipFree(sink);
  (void)hipFree(table);
  if (e != hipSuccess) return qh_fail(QH_EHIP, "random-access benchmark failed: %s", hipGetErrorString(e));
  *ms = total / iters;
  return QH_OK;
}

// lane-operations per second of one kind of LDS operation (see k_lds_rate), chip-wide
extern "C" int qh_bench_lds_rate(int device, int mode, int iters, double* lane_ops_per_s) {
  if (!lane_ops_per_s || mode < 0 || mode > 7 || iters < 1) return qh_fail(QH_EBADARG, "bad argument");
  QH_HIP(hipSetDevice(device));
  unsigned long long* sink = nullptr;
  QH_HIP(hipMalloc((void**)&sink, 8));
  hipEvent_t e0, e1;
  QH_HIP(hipEventCreate(&e0));
  QH_HIP(hipEventCreate(&e1));
  const int blocks = 1024;
  qh_launch_lds_rate(mode, iters, blocks, sink, nullptr);
  QH_HIP(hipEventRecord(e0, nullptr));
  qh_launch_lds_rate(mode, iters, blocks, sink, nullptr);
  QH_HIP(hipEventRecord(e1, nullptr));
  QH_HIP(hipEventSynchronize(e1));
  float ms = 0;
  QH_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(sink);
  *lane_ops_per_s = (double)blocks * 1024.0 * iters * (mode == 7 ? 3.0 : 1.0) / (ms * 1e-3);
  return QH_OK;
}
