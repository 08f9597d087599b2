/*
 * quorum_hip.h -- C ABI of libquorum_hip: Quorum's quality-aware k-mer database
 * build and database_query lookups on AMD MI355X (gfx950).
 *
 * Plain C: opaque handles, pointers and sizes, int status returns, no C++ or
 * torch types.  Every entry point names the upstream Quorum interface it
 * replaces (paths relative to the Quorum source tree).  Upstream reports errors
 * with C++ exceptions / exit(1); here every call returns a qh_status and the
 * message is available from qh_last_error() (thread-local).
 *
 * Value encoding everywhere: v = count << 1 | hq   (src/mer_database.hpp:102-111),
 * 0 = k-mer absent (database_query::operator[] returns (0,0), :284-293).
 * K-mer encoding: jellyfish::mer_dna, A=0 C=1 G=2 T=3, base 0 (the LAST
 * character of the string) in the two low bits, k <= 31.
 *
 * Threading: a build handle is driven by one host thread at a time (the GPU
 * supplies the parallelism that upstream gets from -t threads).  A query handle
 * is immutable after open; its batched lookups may be called concurrently from
 * many host threads, like the const methods of database_query: every call runs
 * on a stream and staging buffers of its own (up to 16 at a time, further
 * callers wait for one).  The table reductions (stats, histogram, dump)
 * serialise on a mutex.
 */
#ifndef QUORUM_HIP_H
#define QUORUM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum qh_status {
  QH_OK = 0,
  QH_EBADARG = 1,  /* invalid argument / unsupported geometry                       */
  QH_EFULL = 2,    /* "Hash is full" (create_database.cc:86-87): cannot grow         */
  QH_EIO = 3,      /* file open/read/write failure                                   */
  QH_EHIP = 4,     /* a HIP runtime call failed                                      */
  QH_EFORMAT = 5,  /* FASTQ chunk is not in strict 4-line form / invalid FASTQ       */
  QH_ENOMEM = 6
} qh_status;

typedef struct qh_build qh_build; /* ~ hash_with_quality + quality_mer_counter */
typedef struct qh_query qh_query; /* ~ database_query                          */

/* One parsed FASTQ record inside a chunk: sequence bytes [seq_off, seq_off+len)
 * and quality bytes [qual_off, qual_off+len) of the same buffer. */
typedef struct qh_record {
  uint32_t seq_off;
  uint32_t len;
  uint32_t qual_off;
  uint32_t reserved;
} qh_record;

typedef struct qh_build_stats {
  uint64_t instances;     /* counted_low  of create_database.cc:58,89  */
  uint64_t hq_instances;  /* counted_high of create_database.cc:58,88  */
  uint64_t distinct;      /* occupied slots                            */
  uint64_t slots;         /* table size of this shard (power of two)   */
  uint64_t records;       /* reads consumed                            */
  uint64_t grows;         /* number of 2x table doublings performed    */
  uint64_t foreign;       /* k-mers dropped because another shard owns them */
  uint64_t window_passes; /* passes over the table by the partitioned build (one per batch of chunks) */
  uint64_t early_preps;   /* FASTQ chunks indexed and cut into segments while their predecessor was still being split */
} qh_build_stats;

const char* qh_last_error(void);
const char* qh_version(void);

/* ------------------------------------------------------------------ build side */

/* hash_with_quality::hash_with_quality(size, key_len=2k, bits, nb_threads, reprobe_limit)
 * (src/mer_database.hpp:78-87) together with the argument rules of
 * create_database main (src/create_database.cc:105-112): k in 1..31, bits in 1..63
 * (where remainder + probe + value exceed one 64-bit slot -- large -b, long mers in small
 * tables -- the values live in a side array of their own like upstream's val_array,
 * mer_database.hpp:81: same content, same file format, direct global-CAS path only), size is
 * rounded up to a power of two like large_hash::array does, min_qual is the
 * threshold compared as a signed char (:48,:81,:110).  device = HIP device id.
 * shard_bits/shard_id: this handle owns shard `shard_id` of 2^shard_bits
 * hash-partitioned shards (0,0 = unsharded). */
int qh_build_create(qh_build** out, int k, int bits, uint64_t size, int min_qual,
                    int reprobe_limit, int device, int shard_bits, int shard_id);
void qh_build_destroy(qh_build* h);

/* Use an existing HIP stream (hipStream_t passed as void*) for all work of this
 * handle; NULL = the handle's own stream.  No upstream counterpart. */
int qh_build_set_stream(qh_build* h, void* hip_stream);

/* quality_mer_counter::start over one chunk of FASTQ text
 * (src/create_database.cc:56-95; the chunk plays the role of the parser jobs of
 * :52,:61-66).  The chunk must begin at a record start and end at a record end.
 * _device: `buf` is device memory already resident in HBM; the call only
 * enqueues kernels.  The host variant copies the chunk to the device first.
 * The GPU fast path accepts strict 4-line records; anything else is detected
 * on the device, nothing of the chunk is inserted, and qh_build_sync() reports
 * QH_EFORMAT so the caller can re-submit through qh_build_add_fastq_host_parsed. */
int qh_build_add_fastq_device(qh_build* h, const void* dev_buf, size_t len);
int qh_build_add_fastq(qh_build* h, const char* buf, size_t len);

/* Same chunk semantics, but records are found by the sequential host parser
 * that follows jellyfish::whole_sequence_parser's FASTQ rules (multi-line
 * sequence/quality accepted; "Truncated fastq file", "wrong number of quals",
 * "header missing" become QH_EFORMAT).  *consumed receives the number of bytes
 * that formed complete records when `final_chunk` is 0 (a trailing partial
 * record is left for the next call); with final_chunk != 0 the whole buffer
 * must parse. */
int qh_build_add_fastq_host_parsed(qh_build* h, const char* buf, size_t len,
                                   int final_chunk, size_t* consumed);

/* Every canonical k-mer of a FASTA buffer ('>' header line, sequence lines up to the next
 * line starting with '>'), each with the quality bit set: the input side of the contaminant
 * set (src/error_correct_reads.cc:86-90 inserts the keys `jellyfish count -C` found). */
int qh_build_add_fasta(qh_build* h, const char* buf, size_t len);

/* Pre-indexed records over a device-resident buffer (record table in host memory). */
int qh_build_add_records_device(qh_build* h, const void* dev_buf, size_t len,
                                const qh_record* recs, size_t nrec);

/* hash_with_quality::add(mer, quality) batched (src/mer_database.hpp:94-113).
 * mers are inserted as given (upstream add() does not canonicalise either).
 * _device takes packed items  mer << 1 | quality  in device memory. */
int qh_build_add_mers(qh_build* h, const uint64_t* mers, const uint8_t* quality, size_t n);
int qh_build_add_mers_device(qh_build* h, const uint64_t* dev_items, size_t n);
/* same for items produced by qh_build_extract_partitioned_device (hashed form) */
int qh_build_add_hashed_device(qh_build* h, const uint64_t* dev_items, size_t n);

/* qh_build_add_mers_device for callers that cannot allocate. NO counterpart in the reference: it exists
 * for code that must not reach hipMalloc (tight serving loops, harnesses that forbid allocation, code
 * around HIP graph capture). Same items, same update rule, same table as qh_build_add_mers_device; the
 * kernel's counters and overflow queue live in a file-scope device array of 512.25 KiB that every GPU
 * the library runs on holds until the process ends, so the call performs no hipMalloc / hipFree.
 * Limits: n <= 2^24 items per call (QH_EBADARG beyond); the table cannot grow here, so an item that hits
 * the reprobe limit makes the call (and the handle) fail with QH_EFULL: size the table at creation.
 * No chunk of another entry point may be in flight (call qh_build_sync first). The call returns when the
 * items are in the table. All calls on one GPU share that GPU's array: calls that overlap on one GPU
 * (from two threads or two handles) race. */
int qh_build_add_mers_device_noalloc(qh_build* h, const uint64_t* dev_items, size_t n);

/* Waits for queued work, performs table doubling when a chunk overflowed
 * (the role of handle_full_ary, src/mer_database.hpp:137-187) and returns the
 * first deferred error: QH_EFULL (growth impossible), QH_EFORMAT, QH_EHIP. */
int qh_build_sync(qh_build* h);

/* hash_with_quality::done (src/mer_database.hpp:128-131): end of input. */
int qh_build_finish(qh_build* h, qh_build_stats* stats);
int qh_build_get_stats(qh_build* h, qh_build_stats* stats);

/* Empty the table and counters (a new database in the same handle; used between benchmark steps). With work in flight
 * on the handle's own stream the reset is noted and applied when that work is collected by the next call, so that a
 * FASTQ chunk handed over next is prepared beside it (qh_build_add_fastq_device); an error of the work in flight is
 * then returned by that next call. No upstream counterpart. */
int qh_build_reset(qh_build* h);

/* Starts the pass over the table for what waits in the window buckets (qh_build_set_defer) and returns without
 * waiting; the next call that collects (qh_build_sync, an add, get_stats, ...) waits for it. qh_build_sync alone does
 * both. No upstream counterpart. */
int qh_build_flush(qh_build* h);

/* hash_with_quality::write(os, &header) (src/mer_database.hpp:115-126) with
 * header.fill_standard()/set_cmdline (src/create_database.cc:100-102). */
int qh_build_write(qh_build* h, const char* path, int argc, const char* const* argv);

/* Sharded serialisation: every shard pwrite()s its slice of the keys and vals
 * images into one file. The shard called with write_header != 0 must run first
 * (it creates the file and sizes it); all shards must have the same table size
 * (see qh_build_grow_to). The calls of different shards must be SERIALISED (one
 * writer at a time, a barrier between ranks in a multi-process build): when the
 * bits+1-wide value fields do not pack evenly into 64-bit words (-b 2, 4, 6 ...),
 * neighbouring shards share a boundary word of the values image, which each
 * writer completes by read-OR-write. No upstream counterpart (one process). */
int qh_build_write_shard(qh_build* h, const char* path, int argc, const char* const* argv,
                         int write_header);

/* The same database with the keys image in the layout of jellyfish::large_hash::array as far as it could be
 * restated without the library (what hash_with_quality::write dumps upstream, src/mer_database.hpp:115-126, and
 * database_query wraps, :273-277): GF(2) matrix position, quadratic reprobes, bit-packed (high key bits, reprobe)
 * entries; header with "matrix1" and "reprobes" and no "qhip" object. PARITY UNPINNED and a known deviation (no
 * "set" bits in straddling words; quorum_amd/csrc/qh_jflayout.h): an upstream reader is NOT expected to decode
 * it. A serialisation switch for the day it can be checked against Jellyfish, self round-trip tested:
 * qh_query_open reads such a file (entries decoded on the host, re-inserted into a native table in HBM).
 * Unsharded handles only; the image is built on the host from the table's content. */
int qh_build_write_jf(qh_build* h, const char* path, int argc, const char* const* argv);
/* Host-only halves of the above (no GPU call): n distinct (mer, count<<1|hq) pairs -> file, and back (entry
 * order; call with cap = 0 and null buffers to size). */
int qh_jf_write_file(const char* path, int k, int bits, uint64_t size, int reprobe_limit, const uint64_t* mers,
                     const uint64_t* vals, uint64_t n, int argc, const char* const* argv);
int qh_jf_read_file(const char* path, int* k, int* bits, uint64_t* size, uint64_t* mers, uint64_t* vals,
                    uint64_t cap, uint64_t* n);

/* Doubles this shard's table until it has at least 2^lsize slots (used to
 * equalise shard sizes before qh_build_write_shard). */
int qh_build_grow_to(qh_build* h, int lsize);

/* The quorum driver's quality-offset detection (src/quorum.in:129-152): the smallest quality
 * character of the first 1000 records (header lines 1..4000) of a FASTQ buffer, minus 2 when it is
 * 35 or 66 (Illumina files without qualities 0 and 1). *min_q_char receives it; QH_EFORMAT with
 * "Found an unusual minimum quality char ..." unless it is 33, 59 or 64, or "Invalid fastq format"
 * when a record stops before its quality line. The driver then runs
 * quorum_create_database -q min_q_char + min_quality (:155-156). Host-only. */
int qh_detect_min_qual_char(const char* buf, size_t len, int* min_q_char);

/* Page-locked host memory for chunk buffers: the host-to-device copy of qh_build_add_fastq /
 * qh_build_add_fastq_host_parsed from such a buffer is a direct DMA (pageable memory goes
 * through a bounce buffer at a fifth of the rate). No counterpart in the reference (its reads
 * stay on the host); used by the quorum_create_database tool. */
int qh_host_alloc(void** out, size_t bytes);
void qh_host_free(void* p);

/* The library's position hash of 2-bit packed mers (host only): the owner shard of a mer is the low
 * shard_bits of it, and the four mers that differ in their first or last base share its low 2k-4 bits.
 * No upstream counterpart (Jellyfish's hash is a GF(2) matrix product); used by the multi-GPU driver
 * and the tests. */
int qh_mer_hash(int k, const uint64_t* mers, size_t n, uint64_t* hashes);

/* Number of leading bytes of a FASTQ buffer that form whole groups of 4 lines:
 * where the CLI cuts a stream into chunks for the GPU fast path. */
size_t qh_fastq_4line_prefix(const char* buf, size_t len, int final_chunk);

/* The n-1 cut points that divide a buffer of whole strict 4-line records (qh_fastq_4line_prefix) into n parts of about
 * equal size, each beginning at a record start: cuts[0] = 0 <= cuts[1] <= ... <= cuts[n] = len. How the multi-GPU build
 * shares a chunk among its GPUs and the corrector tool among its replicas. No upstream counterpart (upstream's threads
 * take jobs of 100 reads from one parser, src/create_database.cc:52,61-66). Host-only. */
int qh_fastq_split_records(const char* buf, size_t len, int n, size_t* cuts);

/* Content export: every (mer, value) of this shard, unordered.  *n receives the
 * number of entries; at most cap are written (call with cap=0 to size). */
int qh_build_dump(qh_build* h, uint64_t* mers, uint64_t* vals, uint64_t cap, uint64_t* n);

/* A query handle over the live table (no file round trip).  The build handle
 * must outlive it and must not be modified while it is in use. */
int qh_build_as_query(qh_build* h, qh_query** out);

/* Multi-GPU exchange support (no upstream counterpart).  Extracts the
 * chunk's (canonical mer, hq) items bucketed by owner shard: items for shard s are
 * written to dev_out[s*cap_per_shard ...] and their number to dev_counts[s]
 * (uint64, device memory, zeroed by the call).  Items are in the library's hashed
 * form  hash(mer) << 1 | hq  (the hash is a bijection of the mer; owner = its low
 * shard_bits): opaque 8-byte words to be exchanged and handed to
 * qh_build_add_hashed_device on the owning shard. */
int qh_build_extract_partitioned_device(qh_build* h, const void* dev_buf, size_t len,
                                        uint64_t* dev_out, uint64_t cap_per_shard,
                                        uint64_t* dev_counts);

/* ------------------------------------------------- several GPUs, one process */

/* The hash-sharded build across n_devices GPUs of one node driven from ONE host process: the multi-GPU form of
 * hash_with_quality + quality_mer_counter (src/create_database.cc:117-123 is one process with -t threads; here one host
 * thread per GPU). `size` is the TOTAL table size; shard d (owner = low log2(n) bits of the position hash) lives on
 * devices[d] (NULL: devices 0..n-1; the same device may be named more than once). n_devices must be a power of two.
 * The k-mers of every chunk are exchanged all-to-all: RCCL (grouped ncclSend/ncclRecv over xGMI, librccl loaded at run
 * time) between distinct devices, plain peer copies otherwise or with QH_MULTI_TRANSPORT=copy. */
typedef struct qh_multi qh_multi;
int qh_multi_create(qh_multi** out, int k, int bits, uint64_t size, int min_qual, int reprobe_limit,
                    int n_devices, const int* devices);
void qh_multi_destroy(qh_multi* m);
/* quality_mer_counter::start (src/create_database.cc:56-95) over one host buffer of FASTQ text. The leading whole groups
 * of 4 lines are cut into n parts at record boundaries, GPU d extracts part d bucketed by owner, the buckets are
 * exchanged and every GPU inserts what it owns (the insert overlaps the next call's extraction). *consumed receives the
 * bytes taken (a partial tail is left for the next call unless final_chunk). Text the GPU indexer rejects (multi-line
 * records) is parsed by the sequential host parser on every shard instead, from then on (Jellyfish's FASTQ rules). */
int qh_multi_add_fastq(qh_multi* m, const char* buf, size_t len, int final_chunk, size_t* consumed);
/* hash_with_quality::done: waits for all shards; `total` sums their statistics (slots = all shards). */
int qh_multi_finish(qh_multi* m, qh_build_stats* total);
/* hash_with_quality::write (src/mer_database.hpp:115-126): equalises the shard sizes and serialises them into one file. */
int qh_multi_write(qh_multi* m, const char* path, int argc, const char* const* argv);
/* Content export of all shards (as qh_build_dump). */
int qh_multi_dump(qh_multi* m, uint64_t* mers, uint64_t* vals, uint64_t cap, uint64_t* n);
/* What the handle is doing: number of shards, 1 if the exchange runs over RCCL, chunks processed, chunks that fell back
 * to the host parser (the GPU indexer rejected one of their parts), bucket overflows resolved by a retry. Any pointer
 * may be NULL. */
int qh_multi_info(const qh_multi* m, int* n_devices, int* uses_rccl, uint64_t* rounds, uint64_t* fallback_chunks,
                  uint64_t* retries);

/* -------------------------------------------------------------- query side */

/* database_header fields (src/mer_database.hpp:43-63 + jellyfish::file_header) of
 * a database file, read without touching a GPU. QH_EFORMAT when the format string
 * is not "binary/quorum_db" (check_format, :60-62). */
typedef struct qh_db_info {
  uint64_t size;         /* slots                         header.size()        */
  uint64_t key_bytes;    /*                               header.key_bytes()   */
  uint64_t value_bytes;  /*                               header.value_bytes() */
  uint64_t offset;       /* start of the keys image       header.offset()      */
  uint32_t key_len;      /* 2k                            header.key_len()     */
  uint32_t val_len;      /* 0                             header.val_len()     */
  uint32_t max_reprobe;  /*                               header.max_reprobe() */
  uint32_t bits;         /* count bits                    header.bits()        */
  uint32_t shard_bits;
  uint32_t window_bits;
  char format[32];
} qh_db_info;
int qh_db_header_info(const char* path, qh_db_info* info);
/* The same from the first bytes of a database held in memory: file_header::read(std::istream&) as
 * database_header uses it (src/mer_database.hpp:257-267). `len` must cover the 9-digit length prefix and the header. */
int qh_db_header_parse(const void* data, size_t len, qh_db_info* info);

/* database_query::database_query(filename, no_mmap) (src/mer_database.hpp:270-278):
 * parses the header and loads the table into HBM of `device`. */
int qh_query_open(qh_query** out, const char* path, int device);
void qh_query_close(qh_query* q);

/* header().key_len()/2, header().bits(), header().size() (src/mer_database.hpp:280,
 * src/error_correct_reads.cc:690, unit_tests/test_mer_database.cc:113-114). */
int qh_query_info(const qh_query* q, int* k, int* bits, uint64_t* size);

/* database_query::operator[] batched (src/mer_database.hpp:284-293). The caller
 * passes canonical mers, as upstream callers do. out[i] = count<<1|hq, 0 absent. */
int qh_query_batch(qh_query* q, const uint64_t* mers, size_t n, uint64_t* out);
int qh_query_batch_device(qh_query* q, const uint64_t* dev_mers, size_t n, uint64_t* dev_out,
                          void* hip_stream);

/* database_query::get_val batched (src/mer_database.hpp:296-299): hq ? count : 0 */
int qh_query_get_val_batch(qh_query* q, const uint64_t* mers, size_t n, uint64_t* out);

/* database_query::get_best_alternatives batched (src/mer_database.hpp:302-329).
 * fmers[i] is the forward mer of a kmer_t; backward[i] = 0 for forward_mer
 * (varies base 0), 1 for backward_mer (varies base k-1, src/kmer.hpp:96-103).
 * counts is [n][4]; ucode/level/nkept are [n] (ucode = 0 when nothing kept,
 * matching the caller-side initialisation at src/error_correct_reads.cc:410). */
int qh_query_best_alternatives_batch(qh_query* q, const uint64_t* fmers, const uint8_t* backward,
                                     size_t n, uint64_t* counts, int32_t* ucode, int32_t* level,
                                     int32_t* nkept);

/* compute_poisson_cutoff__'s table reduction (src/error_correct_reads.cc:650-659):
 * distinct = #{v : (v&1) && v>=2}, total = sum of their counts. */
int qh_query_stats(qh_query* q, uint64_t* distinct, uint64_t* total);

/* poisson_term(lambda, i) (src/error_correct_reads.cc:53-61): e^-lambda lambda^i / i!,
 * Stirling's form from i = 11. */
double qh_poisson_term(double lambda, unsigned i);

/* compute_poisson_cutoff(mer_database.vals(), collision_prob, poisson_threshold)
 * (src/error_correct_reads.cc:650-674; called at :712-715 with apriori_error_rate/3 and
 * poisson_threshold/apriori_error_rate): table reduction on the GPU, then the first
 * x in [2,1000) whose Poisson term falls under the threshold, plus one; 0 if none. */
int qh_query_poisson_cutoff(qh_query* q, double collision_prob, double poisson_threshold,
                            unsigned* cutoff);

/* The contaminant set of src/error_correct_reads.cc:83-99,692-708, built directly from a
 * FASTA file at mer length k (upstream: `jellyfish count -C` dump read back key by key,
 * Makefile.am:54-55): every canonical k-mer of every record. Close with qh_query_close. */
int qh_contaminant_open_fasta(qh_query** out, const char* fasta_path, int k, int device);

/* contaminant_database::is_contaminant (src/error_correct_reads.cc:94) for a batch of
 * canonical mers: flags[i] = 1 iff mers[i] is a key of `set`. */
int qh_query_is_member_batch(qh_query* set, const uint64_t* mers, size_t n, uint8_t* flags);

/* The database lookup and the contaminant test the corrector makes on the same canonical
 * mer (src/error_correct_reads.cc:348,390,620) in one pass: out[i] as qh_query_batch (may be
 * NULL), flags[i] as qh_query_is_member_batch. Both handles must have the same k
 * (:703-705) and device. */
int qh_query_batch_contaminant(qh_query* q, qh_query* contaminant, const uint64_t* mers, size_t n,
                               uint64_t* out, uint8_t* flags);
int qh_query_batch_contaminant_device(qh_query* q, qh_query* contaminant, const uint64_t* dev_mers,
                                      size_t n, uint64_t* dev_out, uint8_t* dev_flags,
                                      void* hip_stream);

/* histo_mer_database (src/histo_mer_database.cc:13-21): histo[min(count,1000)][hq] */
int qh_query_histo(qh_query* q, uint64_t* histo /* [1001][2] */);

/* database_query::begin()/end() iteration (src/mer_database.hpp:331-361) as a
 * bulk export of (mer, value) pairs. */
int qh_query_dump(qh_query* q, uint64_t* mers, uint64_t* vals, uint64_t cap, uint64_t* n);

/* ------------------------------------------------------ read corrector */

/* The switches of quorum_error_correct_reads (src/error_correct_reads_cmdline.yaggo:6-71)
 * as main() hands them to error_correct_t (src/error_correct_reads.cc:720-736). */
typedef struct qh_correct_params {
  int32_t skip;             /* -s  bases to skip for the start k-mer            (1) */
  int32_t good;             /* -g  good k-mers in a row for an anchor           (2) */
  int32_t anchor;           /* -a  minimum count of an anchor k-mer             (3) */
  int32_t min_count;        /* -m  minimum count of a "good" k-mer              (1) */
  int32_t cutoff;           /* -p  Poisson cutoff (qh_query_poisson_cutoff when not given, :712-718) */
  int32_t window;           /* -w  window size, 0 = k   (error_correct_reads.cc:206; yaggo default 10) */
  int32_t error;            /* -e  errors allowed in a window, 0 = k/2 (:207; yaggo default 3)        */
  int32_t qual_cutoff;      /* -q/-Q compared as a char; 127 when neither is given (:684-685)          */
  int32_t trim_contaminant; /* --trim-contaminant                                                      */
  int32_t do_homo_trim;     /* --homo-trim given                                                       */
  int32_t homo_trim;        /* its value                                                               */
  int32_t no_discard;       /* -d  output a single N for a discarded read                              */
  double collision_prob;    /* apriori_error_rate / 3 (:734)                                           */
  double poisson_threshold; /* --poisson-threshold (:735)                                              */
} qh_correct_params;

typedef struct qh_correct_stats {
  uint64_t reads;          /* records seen                                 */
  uint64_t corrected;      /* reads written to the .fa stream with a sequence */
  uint64_t skipped;        /* "Skipped <header>: <reason>" lines           */
  uint64_t substitutions;  /* pos:sub:X-Y entries                          */
  uint64_t truncations;    /* pos:3_trunc + pos:5_trunc entries            */
  uint64_t bases_in;       /* sequence bytes of the chunk                  */
  uint64_t bases_out;      /* sequence bytes written                       */
} qh_correct_stats;

typedef struct qh_corrector qh_corrector; /* ~ error_correct_t + its error_correct_instance workers */

/* error_correct_t set up as in main (src/error_correct_reads.cc:720-736) over an open
 * database and an optional contaminant set (NULL = contaminant_no_database, :75-81; same k
 * and device as db, :703-705). Both handles must outlive the corrector. The corrector may keep a
 * private copy of the database re-hashed to a lower load for its walks (QH_EC_TABLE_LOAD, default 0.3,
 * 0 = never; only when the copy takes less than a quarter of the free device memory). */
int qh_corrector_create(qh_corrector** out, qh_query* db, qh_query* contaminant, const qh_correct_params* p);
void qh_corrector_destroy(qh_corrector* c);

/* error_correct_instance::start (src/error_correct_reads.cc:246-341) over one chunk of FASTQ
 * text resident in HBM (16-byte aligned, strict 4-line records, whole records only): the
 * record index, find_starting_mer (:609-643), extend in both directions (:386-565), homo_trim
 * (:567-597) and the text of both output streams are made on the GPU; the call only enqueues
 * kernels on the corrector's stream (or `hip_stream` when not NULL). One read per lane; reads
 * keep their input order in the output (upstream with -t 1). QH_EFORMAT is deferred to
 * qh_corrector_fetch, like qh_build_add_fastq_device defers it to qh_build_sync. */
int qh_corrector_correct_device(qh_corrector* c, const void* dev_buf, size_t len, void* hip_stream);

/* Waits for the chunk and hands out its two text streams: `fa` is what upstream writes to
 * <prefix>.fa (">header fwd_log bwd_log\nSEQUENCE\n", :334-336; ">header\nN\n" for a discarded
 * read under -d, :274-275), `log` what it writes to <prefix>.log ("Skipped header: reason\n",
 * :272). The pointers are library-owned pinned host memory; fetches alternate between two buffers,
 * so they stay valid through the NEXT correct + fetch on this corrector and die with the one after
 * (a tool writes chunk i out while chunk i+1 is corrected). stats may be NULL. */
int qh_corrector_fetch(qh_corrector* c, const char** fa, size_t* fa_len, const char** log, size_t* log_len,
                       qh_correct_stats* stats);

/* Host-buffer form: copies the chunk to the device and corrects it. The GPU path takes the
 * leading whole 4-line groups; a chunk the GPU indexer rejects is parsed by the sequential
 * host parser with Jellyfish's FASTQ rules (multi-line records) and corrected by the same
 * kernel. *consumed receives the bytes that formed complete records when final_chunk is 0.
 * Results are collected with qh_corrector_fetch. */
int qh_corrector_correct_fastq(qh_corrector* c, const char* buf, size_t len, int final_chunk, size_t* consumed);

/* HIP-event time of the correction kernel (k_correct_reads) and of the whole enqueued chunk
 * (index + correct + format) of the last qh_corrector_correct_* call, in milliseconds. */
int qh_corrector_kernel_time(qh_corrector* c, double* correct_ms, double* chunk_ms);

/* -------------------------------------------------- measurement support */

/* HIP-event timing of the main kernels of each chunk, recorded on the stream the
 * kernels are launched on. which: 0 = direct global-CAS build / item insert,
 * 1 = partition level 1 (includes the k-mer extraction), 2 = partition level 2,
 * 3 = LDS window build. kernel_time returns the accumulated milliseconds and the
 * number of timed launches since enabling. No upstream counterpart (upstream
 * computes counted_high/low and never prints them, create_database.cc:58,88-89). */
int qh_build_enable_timing(qh_build* h, int on);
int qh_build_kernel_time(qh_build* h, int which, double* total_ms, uint64_t* launches);

/* Build strategy: 0 = automatic, 1 = direct (every k-mer is a global 64-bit
 * atomicCAS), 2 = radix-partitioned (k-mers are bucketed by probe window and
 * applied in LDS). level1_bits >= 0 forces the fan-out of the first partition
 * level (-1 = automatic). Both strategies produce the same table format and the
 * same content. Also settable with QH_BUILD_MODE=direct|partition, QH_LEVEL1_BITS. */
int qh_build_set_mode(qh_build* h, int mode, int level1_bits);

/* Deferred window pass of the radix-partitioned build. A partitioned chunk ends in one pass over the whole table
 * (every probe window is loaded into LDS, its bucket applied, the window streamed back): 16 bytes per SLOT whatever
 * the chunk's size, which dominates when chunks are small beside the table (a 2^31-slot table: 34 GB per chunk). With
 * a budget the window buckets of successive chunks accumulate in HBM instead and the table is streamed once per
 * BATCH: when the budget is used up, and whenever the table must be current -- qh_build_sync, finish, get_stats, dump,
 * as_query, write, grow_to and the direct-path entry points flush first, so results never depend on the budget.
 * max_bytes: memory the buckets of a batch may take; 0 = every chunk is flushed at once (the behaviour until round 3),
 * negative = automatic (three times the table, at most half of the free device memory; the default). Also
 * QH_DEFER_BYTES. No upstream counterpart (upstream updates its table k-mer by k-mer). */
int qh_build_set_defer(qh_build* h, long long max_bytes);

/* Random-access roofline microbenchmark: n_updates uniformly random
 * 64-bit atomicCAS (mode 0) or 8-byte loads (mode 1) over a table of 2^lsize
 * words on `device`; returns the average kernel time of `iters` launches in
 * milliseconds through *ms. */
int qh_bench_random_access(int device, int lsize, uint64_t n_updates, int mode, int iters,
                           double* ms);

/* LDS operation-rate microbenchmark (design input for the LDS window build): lane
 * operations per second chip-wide for mode 0 ds_read_b64, 1 ds_write_b64,
 * 2 ds_cmpst_rtn_b64, 3 ds_cmpst_rtn_b32, 4 ds_add_rtn_u32, 5 ds_add_u32,
 * 6 ds_add_rtn_u64, 7 read+write+read of one slot. */
int qh_bench_lds_rate(int device, int mode, int iters, double* lane_ops_per_s);

#ifdef __cplusplus
}
#endif
#endif /* QUORUM_HIP_H */
