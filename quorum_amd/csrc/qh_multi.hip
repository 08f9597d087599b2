// qh_multi.hip -- the hash-sharded build across several GPUs of one node from ONE host process (C++ host, as the
// reference is one process: src/create_database.cc:117-123 builds one hash_with_quality with -t threads). SURVEY 8(e):
// every GPU takes a share of the reads, extracts their k-mers bucketed by owner shard (owner = low bits of the position
// hash), the buckets are exchanged all-to-all over xGMI, and every GPU inserts what it owns into its shard of the table.
//
// One host thread per GPU drives its device; a chunk of reads is one ROUND:
//     stage + extract by owner (all GPUs at once)  |barrier|  exchange: every GPU pulls its bucket from every GPU
//     |barrier|  insert what arrived (asynchronous: it overlaps the next round's staging and extraction)
// Transport of the exchange: RCCL (grouped ncclSend / ncclRecv, one communicator per GPU from ncclCommInitAll; librccl is
// loaded with dlopen so that the library carries no link-time dependency and does not clash with the copy of RCCL that
// PyTorch brings) or plain peer copies (hipMemcpyPeerAsync: in one process an all-to-all over xGMI is N x (N-1) peer
// copies; also what runs when several shards share one device, which is how the one-GPU test box exercises this file).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "qh_host.h"
#include "qh_internal.h"

namespace {

// the few RCCL entry points the exchange needs, resolved at run time
struct rccl_api {
  void* lib = nullptr;
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t s) = nullptr;
  int (*Recv)(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t s) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool load() {
    lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return false;
    CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
    Send = (decltype(Send))dlsym(lib, "ncclSend");
    Recv = (decltype(Recv))dlsym(lib, "ncclRecv");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv;
  }
};
// largest single RCCL message in items (512 MiB); QH_EXCHANGE_MAX_ITEMS lowers or raises it (tests)
uint64_t max_message_items() {
  const char* e = getenv("QH_EXCHANGE_MAX_ITEMS");
  const long long v = e ? atoll(e) : 0;
  return v > 0 ? (uint64_t)v : (1ull << 26);
}
const int kNcclUint64 = 5;  // ncclUint64 (rccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4, ncclUint64 5)

class round_barrier {
  std::mutex mu;
  std::condition_variable cv;
  int n, waiting = 0;
  unsigned long gen = 0;
public:
  explicit round_barrier(int n_) : n(n_) {}
  // on_last runs in the last thread to arrive, before anyone is released: where the threads agree on what to do next
  template <typename F>
  void wait(F on_last) {
    std::unique_lock<std::mutex> lk(mu);
    const unsigned long g = gen;
    if (++waiting == n) { on_last(); waiting = 0; ++gen; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g; });
  }
};

struct shard_dev {
  int device = 0;
  qh_build* h = nullptr;    // the shard
  qh_build* hx = nullptr;   // extraction handle (its table is never used)
  hipStream_t xs = nullptr; // stream of staging and of the exchange this shard receives through
  uint8_t* d_chunk = nullptr; size_t chunk_cap = 0;
  uint64_t* d_buckets = nullptr; size_t bucket_cap = 0;  // n shards x bucket_cap items
  uint64_t* d_counts = nullptr;
  uint64_t* d_recv[2] = {nullptr, nullptr}; size_t recv_cap[2] = {0, 0};
  void* comm = nullptr;
  int rc = QH_OK;
  std::string err;
};

}  // namespace

struct qh_multi {
  int n = 0, shard_bits = 0, k = 0;
  bool use_rccl = false;
  rccl_api nccl;
  std::vector<shard_dev> sh;
  std::vector<uint64_t> counts;  // counts[s * n + d]: items of shard d extracted on shard s in this round
  uint64_t rounds = 0;
  uint64_t fallback_chunks = 0;  // chunks the GPU indexer rejected: every shard scanned them whole (host parser)
  bool slow_mode = false;        // a chunk was rejected: the rest of the input goes to the host parser directly
  uint64_t retries = 0;          // bucket overflows resolved by a larger capacity
  std::mutex stat_mu;
};

static int dev_ensure(void** p, size_t* cap, size_t need, const char* what) {
  if (*p && *cap >= need) return QH_OK;
  if (*p) { hipError_t e = hipFree(*p); (void)e; *p = nullptr; *cap = 0; }
  const size_t want = need + need / 8 + 256;
  hipError_t e = hipMalloc(p, want);
  if (e != hipSuccess) { *p = nullptr; (void)hipGetLastError(); return qh_fail(QH_ENOMEM, "hipMalloc of %zu bytes for %s failed: %s", want, what, hipGetErrorString(e)); }
  *cap = want;
  return QH_OK;
}

extern "C" void qh_multi_destroy(qh_multi* m) {
  if (!m) return;
  for (auto& s : m->sh) {
    hipError_t e = hipSetDevice(s.device); (void)e;
    if (s.xs) { e = hipStreamSynchronize(s.xs); (void)e; }
    if (s.comm && m->nccl.CommDestroy) m->nccl.CommDestroy(s.comm);
    if (s.h) qh_build_destroy(s.h);
    if (s.hx) qh_build_destroy(s.hx);
    void* ptrs[] = {s.d_chunk, s.d_buckets, s.d_counts, s.d_recv[0], s.d_recv[1]};
    for (void* p : ptrs) if (p) { e = hipFree(p); (void)e; }
    if (s.xs) { e = hipStreamDestroy(s.xs); (void)e; }
  }
  if (m->nccl.lib) dlclose(m->nccl.lib);
  delete m;
}

extern "C" int qh_multi_create(qh_multi** out, int k, int bits, uint64_t size, int min_qual, int reprobe_limit,
                               int n_devices, const int* devices) {
  if (!out) return qh_fail(QH_EBADARG, "null output handle");
  *out = nullptr;
  if (n_devices < 1 || n_devices > 256 || (n_devices & (n_devices - 1)))
    return qh_fail(QH_EBADARG, "the number of GPUs must be a power of two (got %d)", n_devices);
  int ndev = 0;
  QH_HIP(hipGetDeviceCount(&ndev));
  qh_multi* m = new qh_multi();
  m->n = n_devices;
  m->k = k;
  while ((1 << m->shard_bits) < n_devices) ++m->shard_bits;
  m->sh.resize(n_devices);
  m->counts.assign((size_t)n_devices * n_devices, 0);
  bool distinct = true;
  for (int d = 0; d < n_devices; ++d) {
    m->sh[d].device = devices ? devices[d] : d;
    if (m->sh[d].device < 0 || m->sh[d].device >= ndev) {
      const int bad = m->sh[d].device;
      qh_multi_destroy(m);
      return qh_fail(QH_EBADARG, "HIP device %d does not exist (%d visible)", bad, ndev);
    }
    for (int e = 0; e < d; ++e) if (m->sh[e].device == m->sh[d].device) distinct = false;
  }
  for (int d = 0; d < n_devices; ++d) {
    shard_dev& s = m->sh[d];
    int rc = qh_build_create(&s.h, k, bits, size, min_qual, reprobe_limit, s.device, m->shard_bits, d);
    if (rc == QH_OK) rc = qh_build_create(&s.hx, k, bits, (uint64_t)n_devices << QH_WINDOW_BITS, min_qual, reprobe_limit, s.device, m->shard_bits, d);
    if (rc == QH_OK) {
      hipError_t e = hipSetDevice(s.device);
      if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.xs, hipStreamNonBlocking);
      if (e == hipSuccess) e = hipMalloc((void**)&s.d_counts, sizeof(uint64_t) * n_devices);
      if (e != hipSuccess) rc = qh_fail(QH_EHIP, "setting up device %d failed: %s", s.device, hipGetErrorString(e));
    }
    if (rc != QH_OK) { const std::string keep = qh_last_error(); qh_multi_destroy(m); return qh_fail(rc, "%s", keep.c_str()); }
  }
  // transport: RCCL between distinct devices unless QH_MULTI_TRANSPORT=copy; shards that share a device can only copy
  const char* tr = getenv("QH_MULTI_TRANSPORT");
  // (one device with QH_MULTI_TRANSPORT=rccl: the communicator, the grouped send / receive and the rounds run with
  // the shard sending to itself -- the rehearsal a one-GPU box allows)
  const bool force_rccl = tr && !strcmp(tr, "rccl");
  const bool want_rccl = (n_devices > 1 || force_rccl) && distinct && !(tr && !strcmp(tr, "copy"));
  if (want_rccl && m->nccl.load()) {
    std::vector<void*> comms(n_devices, nullptr);
    std::vector<int> devs(n_devices);
    for (int d = 0; d < n_devices; ++d) devs[d] = m->sh[d].device;
    const int nrc = m->nccl.CommInitAll(comms.data(), n_devices, devs.data());
    if (nrc == 0) {
      for (int d = 0; d < n_devices; ++d) m->sh[d].comm = comms[d];
      m->use_rccl = true;
    } else if (tr && !strcmp(tr, "rccl")) {
      const std::string why = m->nccl.GetErrorString ? m->nccl.GetErrorString(nrc) : "?";
      qh_multi_destroy(m);
      return qh_fail(QH_EHIP, "ncclCommInitAll failed: %s", why.c_str());
    }
  } else if (tr && !strcmp(tr, "rccl")) {
    qh_multi_destroy(m);
    return qh_fail(QH_EBADARG, "QH_MULTI_TRANSPORT=rccl needs distinct devices and a loadable librccl.so");
  }
  if (!m->use_rccl && distinct && n_devices > 1) {  // peer copies: let every device reach every other (best effort)
    for (int d = 0; d < n_devices; ++d) {
      hipError_t e = hipSetDevice(m->sh[d].device); (void)e;
      for (int s = 0; s < n_devices; ++s)
        if (s != d) { e = hipDeviceEnablePeerAccess(m->sh[s].device, 0); (void)e; (void)hipGetLastError(); }
    }
  }
  *out = m;
  return QH_OK;
}

extern "C" int qh_multi_info(const qh_multi* m, int* n_devices, int* uses_rccl, uint64_t* rounds, uint64_t* fallback_chunks,
                             uint64_t* retries) {
  if (!m) return qh_fail(QH_EBADARG, "null handle");
  if (n_devices) *n_devices = m->n;
  if (uses_rccl) *uses_rccl = m->use_rccl ? 1 : 0;
  if (rounds) *rounds = m->rounds;
  if (fallback_chunks) *fallback_chunks = m->fallback_chunks;
  if (retries) *retries = m->retries;
  return QH_OK;
}

// The first record start at or after `from` in a chunk of strict 4-line records, found from the next few lines alone:
// a line that starts with '@' and whose second successor starts with '+' is a header and nothing else (see
// qh_fastq_4line_prefix). Returns false when eight lines do not settle it.
static bool record_start_after(const char* buf, size_t len, size_t from, size_t* out) {
  enum { LOOK = 8 };
  size_t st[LOOK + 1];
  int n = 0;
  size_t p = from;
  if (p > 0) {  // into the line that is under way: its successor is the first line seen whole
    const char* q = p < len ? (const char*)memchr(buf + p - 1, '\n', len - (p - 1)) : nullptr;
    if (!q) { *out = len; return true; }
    p = (size_t)(q - buf) + 1;
  }
  bool reached_end = false;
  while (n <= LOOK) {
    if (p >= len) { reached_end = true; break; }
    st[n++] = p;
    const char* q = (const char*)memchr(buf + p, '\n', len - p);
    if (!q) { reached_end = true; break; }
    p = (size_t)(q - buf) + 1;
  }
  for (int i = 0; i + 2 < n; ++i)
    if (buf[st[i]] == '@' && buf[st[i + 2]] == '+') { *out = st[i]; return true; }
  if (reached_end) { *out = len; return true; }  // the tail of the chunk's last record: it stays with the previous part
  return false;
}

// the n-1 cut points of a chunk of whole 4-line records: part d = [cut[d], cut[d+1]) begins at a record start
static void split_records(const char* buf, size_t len, int n, std::vector<size_t>& cut) {
  cut.assign(n + 1, 0);
  cut[n] = len;
  if (n == 1) return;
  {
    bool ok = true;
    for (int d = 1; d < n && ok; ++d) ok = record_start_after(buf, len, len * d / n, &cut[d]);
    if (ok) {
      for (int d = 1; d <= n; ++d) if (cut[d] < cut[d - 1]) cut[d] = cut[d - 1];
      return;
    }
    cut.assign(n + 1, 0);
    cut[n] = len;
  }
  // (lines too long for the look-ahead) newlines per slice, counted in parallel; the record phase of a slice's first
  // line follows from the prefix
  std::vector<size_t> nl(n, 0);
  auto count = [&](int d) {
    const size_t a = len * d / n, b = len * (d + 1) / n;
    size_t c = 0;
    const char* p = buf + a;
    const char* e = buf + b;
    while (p < e && (p = (const char*)memchr(p, '\n', e - p))) { ++c; ++p; }
    nl[d] = c;
  };
  std::vector<std::thread> th;
  for (int d = 1; d < n; ++d) th.emplace_back(count, d);
  count(0);
  for (auto& t : th) t.join();
  size_t lines = 0;  // complete lines before the slice
  for (int d = 1; d < n; ++d) {
    lines += nl[d - 1];
    // advance from the slice start to the next line whose number is a multiple of 4
    size_t p = len * d / n, ln = lines;
    // the slice may start inside a line: that line is number `ln` and ends at the next newline
    for (;;) {
      const char* q = p < len ? (const char*)memchr(buf + p, '\n', len - p) : nullptr;
      if (!q) { p = len; break; }
      p = (size_t)(q - buf) + 1;
      if ((++ln & 3) == 0) break;
    }
    cut[d] = p < cut[d - 1] ? cut[d - 1] : p;
  }
  for (int d = 1; d <= n; ++d) if (cut[d] < cut[d - 1]) cut[d] = cut[d - 1];
}

extern "C" int qh_fastq_split_records(const char* buf, size_t len, int n, size_t* cuts) {
  if (!buf || !cuts || n < 1) return qh_fail(QH_EBADARG, "bad argument");
  std::vector<size_t> cut;
  split_records(buf, len, n, cut);
  for (int d = 0; d <= n; ++d) cuts[d] = cut[d];
  return QH_OK;
}

// One round over a host buffer of FASTQ text (pinned memory copies fastest). The leading whole groups of 4 lines go
// through the GPU path; *consumed receives how many bytes were taken (a partial tail stays for the next call unless
// final_chunk). If the GPU indexer rejects the text (multi-line records), nothing of the round is inserted and the
// buffer goes through the sequential host parser on every shard instead, each keeping the k-mers it owns.
extern "C" int qh_multi_add_fastq(qh_multi* m, const char* buf, size_t len, int final_chunk, size_t* consumed) {
  if (!m) return qh_fail(QH_EBADARG, "null handle");
  if (consumed) *consumed = 0;
  if (len == 0) return QH_OK;
  if (!buf) return qh_fail(QH_EBADARG, "null chunk");
  const int n = m->n;
  const size_t whole_len = len;
  if (!m->slow_mode) {
    len = qh_fastq_4line_prefix(buf, len, final_chunk);
    if (len == 0) {
      if (!final_chunk) return QH_OK;   // not even one group of four lines yet: the caller reads on
      m->slow_mode = true;
    }
  }
  std::vector<size_t> cut;
  if (m->slow_mode) cut.assign(n + 1, 0);   // nothing for the GPU path
  else split_records(buf, len, n, cut);
  size_t slow_used = 0;
  round_barrier bar(n);
  const int slot = (int)(m->rounds & 1);
  std::vector<char> need_fallback(n, 0);
  auto failed_any = [&]() { for (auto& s : m->sh) if (s.rc != QH_OK) return true; return false; };
  auto fail_here = [&](shard_dev& s, int rc) { s.rc = rc; s.err = qh_last_error(); };

  // what every thread does next is decided once per phase, by the last thread to reach the barrier: a shard that failed
  // (out of memory for its receive buffer, say) must keep ALL shards out of the grouped send / receive, or its peers
  // would wait for it for ever
  bool bad_extract = false, bad_alloc = false, bad_exchange = false, slow_all = false;

  auto worker = [&](int d) {
    shard_dev& s = m->sh[d];
    const size_t off = cut[d], plen = cut[d + 1] - cut[d];
    const uint64_t records_before = s.hx->st.records;
    // ---- stage + extract by owner
    do {
      if (hipSetDevice(s.device) != hipSuccess) { qh_fail(QH_EHIP, "hipSetDevice(%d) failed", s.device); fail_here(s, QH_EHIP); break; }
      for (int t = 0; t < n; ++t) m->counts[(size_t)d * n + t] = 0;
      if (plen == 0) break;
      int rc = dev_ensure((void**)&s.d_chunk, &s.chunk_cap, ((plen + 15) & ~(size_t)15) + 16, "chunk staging");
      if (rc) { fail_here(s, rc); break; }
      hipError_t e = hipMemcpyAsync(s.d_chunk, buf + off, plen, hipMemcpyHostToDevice, s.xs);
      if (e == hipSuccess) e = hipStreamSynchronize(s.xs);
      if (e != hipSuccess) { qh_fail(QH_EHIP, "copying a chunk to device %d failed: %s", s.device, hipGetErrorString(e)); fail_here(s, QH_EHIP); break; }
      size_t cap = (plen / 2 / n) + (plen / 2 / n) / 4 + 4096;
      for (;;) {
        if (s.bucket_cap < cap) {
          size_t bytes = 0;
          if (s.d_buckets) { hipError_t x = hipFree(s.d_buckets); (void)x; s.d_buckets = nullptr; }
          rc = dev_ensure((void**)&s.d_buckets, &bytes, (size_t)n * cap * 8, "owner buckets");
          if (rc) break;
          s.bucket_cap = cap;
        }
        rc = qh_build_extract_partitioned_device(s.hx, s.d_chunk, plen, s.d_buckets, s.bucket_cap, s.d_counts);
        if (rc == QH_EBADARG && strstr(qh_last_error(), "cap_per_shard too small")) {  // a skewed chunk: again with room to spare
          cap = s.bucket_cap * 2;
          std::lock_guard<std::mutex> lk(m->stat_mu);
          ++m->retries;
          continue;
        }
        break;
      }
      if (rc == QH_EFORMAT) { need_fallback[d] = 1; break; }  // not strict 4-line FASTQ: every shard scans this part whole
      if (rc) { fail_here(s, rc); break; }
      e = hipMemcpy(&m->counts[(size_t)d * n], s.d_counts, sizeof(uint64_t) * n, hipMemcpyDeviceToHost);
      if (e != hipSuccess) { qh_fail(QH_EHIP, "reading the bucket counts failed: %s", hipGetErrorString(e)); fail_here(s, QH_EHIP); }
    } while (0);
    bar.wait([&] {
      bad_extract = failed_any();
      slow_all = m->slow_mode;
      for (int t = 0; t < n; ++t) slow_all |= need_fallback[t] != 0;
    });
    const bool slow = slow_all;
    // the text goes to the host parser on every shard: what the parts that did pass the GPU indexer counted is dropped
    if (slow) s.hx->st.records = records_before;
    // ---- receive buffer first, so that a shard that cannot have one is known before anybody sends
    uint64_t total = 0;
    if (!bad_extract && !slow) {
      for (int t = 0; t < n; ++t) total += m->counts[(size_t)t * n + d];
      int rc = dev_ensure((void**)&s.d_recv[slot], &s.recv_cap[slot], (total + 1) * 8, "receive buffer");
      if (rc) fail_here(s, rc);
    }
    bar.wait([&] { bad_alloc = failed_any(); });
    // ---- exchange: this shard's bucket from every device into its receive buffer
    if (!bad_alloc && !slow) {
      do {
        // the insert of two rounds ago read this buffer: it completed before the previous round's insert was queued
        hipError_t e = hipSuccess;
        if (m->use_rccl) {
          // Messages are capped (kMaxMessageItems): this image's RCCL delivered only the first half of a 1.6 GB message,
          // without an error (tools/rccl_self_check.py; up to 1 GiB arrives whole). Round r carries items
          // [r * cap, (r + 1) * cap) of every message; all threads see the same count matrix, hence the same rounds.
          uint64_t largest = 0;
          for (size_t i = 0; i < (size_t)n * n; ++i) largest = std::max<uint64_t>(largest, m->counts[i]);
          const uint64_t cap_items = max_message_items();
          const uint64_t rounds = std::max<uint64_t>(1, (largest + cap_items - 1) / cap_items);
          int nrc = 0;
          for (uint64_t r = 0; r < rounds && nrc == 0; ++r) {
            nrc = m->nccl.GroupStart();
            uint64_t at = 0;
            for (int t = 0; t < n && nrc == 0; ++t) {
              const uint64_t out_n = m->counts[(size_t)d * n + t], in_n = m->counts[(size_t)t * n + d];
              const uint64_t olo = std::min(out_n, r * cap_items), ohi = std::min(out_n, (r + 1) * cap_items);
              const uint64_t ilo = std::min(in_n, r * cap_items), ihi = std::min(in_n, (r + 1) * cap_items);
              if (ohi > olo) nrc = m->nccl.Send(s.d_buckets + (size_t)t * s.bucket_cap + olo, ohi - olo, kNcclUint64, t, s.comm, s.xs);
              if (nrc == 0 && ihi > ilo) nrc = m->nccl.Recv(s.d_recv[slot] + at + ilo, ihi - ilo, kNcclUint64, t, s.comm, s.xs);
              at += in_n;
            }
            const int erc = m->nccl.GroupEnd();
            if (nrc == 0) nrc = erc;
          }
          if (nrc != 0) { qh_fail(QH_EHIP, "RCCL exchange failed: %s", m->nccl.GetErrorString ? m->nccl.GetErrorString(nrc) : "?"); fail_here(s, QH_EHIP); break; }
        } else {
          uint64_t at = 0;
          for (int t = 0; t < n && e == hipSuccess; ++t) {
            const uint64_t in_n = m->counts[(size_t)t * n + d];
            const shard_dev& from = m->sh[t];
            if (in_n) {
              const uint64_t* src = from.d_buckets + (size_t)d * from.bucket_cap;
              e = from.device == s.device ? hipMemcpyAsync(s.d_recv[slot] + at, src, in_n * 8, hipMemcpyDeviceToDevice, s.xs)
                                          : hipMemcpyPeerAsync(s.d_recv[slot] + at, s.device, src, from.device, in_n * 8, s.xs);
            }
            at += in_n;
          }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s.xs);
        if (e != hipSuccess) { qh_fail(QH_EHIP, "exchange into device %d failed: %s", s.device, hipGetErrorString(e)); fail_here(s, QH_EHIP); }
      } while (0);
    }
    bar.wait([&] { bad_exchange = failed_any(); });  // every bucket has been read: the next round may overwrite them
    // ---- insert what arrived; returns once it is queued (it overlaps the next round's staging and extraction)
    if (!bad_exchange) {
      int rc;
      if (!slow) {
        rc = total ? qh_build_add_hashed_device(s.h, s.d_recv[slot], total) : qh_build_sync(s.h);
      } else {
        // the sequential host parser (Jellyfish's FASTQ rules) over the whole buffer on EVERY shard, each keeping what it owns
        size_t used = 0;
        rc = qh_build_add_fastq_host_parsed(s.h, buf, whole_len, final_chunk, &used);
        if (d == 0) slow_used = used;
      }
      if (rc) fail_here(s, rc);
    }
  };
  std::vector<std::thread> th;
  for (int d = 1; d < n; ++d) th.emplace_back(worker, d);
  worker(0);
  for (auto& t : th) t.join();
  ++m->rounds;
  bool slow = m->slow_mode;
  for (int d = 0; d < n; ++d) slow |= need_fallback[d] != 0;
  if (slow) { ++m->fallback_chunks; m->slow_mode = true; }
  for (auto& s : m->sh)
    if (s.rc != QH_OK) return qh_fail(s.rc, "%s", s.err.c_str());
  if (consumed) *consumed = slow ? slow_used : len;
  return QH_OK;
}

extern "C" int qh_multi_finish(qh_multi* m, qh_build_stats* total) {
  if (!m) return qh_fail(QH_EBADARG, "null handle");
  qh_build_stats t;
  memset(&t, 0, sizeof t);
  for (auto& s : m->sh) {
    qh_build_stats st, sx;
    int rc = qh_build_finish(s.h, &st);
    if (rc) return rc;
    if ((rc = qh_build_get_stats(s.hx, &sx))) return rc;
    t.instances += st.instances; t.hq_instances += st.hq_instances; t.distinct += st.distinct;
    t.slots += st.slots; t.grows += st.grows; t.window_passes += st.window_passes; t.early_preps += st.early_preps;
    t.records += sx.records;   // counted where the reads were extracted
  }
  // reads that went through the host parser were parsed by every shard: count them once (shard 0's figure)
  qh_build_stats s0;
  if (qh_build_get_stats(m->sh[0].h, &s0) == QH_OK) t.records += s0.records;
  if (total) *total = t;
  return QH_OK;
}

// every shard's (mer, value) pairs, shard after shard (content export for tests and diff tools)
extern "C" int qh_multi_dump(qh_multi* m, uint64_t* mers, uint64_t* vals, uint64_t cap, uint64_t* n) {
  if (!m || !n) return qh_fail(QH_EBADARG, "null argument");
  uint64_t at = 0;
  for (auto& s : m->sh) {
    uint64_t got = 0;
    const uint64_t room = cap > at ? cap - at : 0;
    int rc = qh_build_dump(s.h, room ? mers + at : nullptr, room ? vals + at : nullptr, room, &got);
    if (rc) return rc;
    at += got;
  }
  *n = at;
  return QH_OK;
}

// hash_with_quality::write for the sharded table: the shards are brought to one size (a shard that doubled after an
// overflow decides it) and write their slices of the one file one after the other (boundary words of the values image
// are shared between neighbouring shards: single writer at a time).
extern "C" int qh_multi_write(qh_multi* m, const char* path, int argc, const char* const* argv) {
  if (!m || !path) return qh_fail(QH_EBADARG, "null argument");
  uint64_t slots = 0;
  for (auto& s : m->sh) {
    qh_build_stats st;
    int rc = qh_build_finish(s.h, &st);
    if (rc) return rc;
    if (st.slots > slots) slots = st.slots;
  }
  int lsize = 0;
  while ((1ull << lsize) < slots) ++lsize;
  for (auto& s : m->sh) { int rc = qh_build_grow_to(s.h, lsize); if (rc) return rc; }
  for (int d = 0; d < m->n; ++d) {
    int rc = qh_build_write_shard(m->sh[d].h, path, argc, argv, d == 0 ? 1 : 0);
    if (rc) return rc;
  }
  return QH_OK;
}
