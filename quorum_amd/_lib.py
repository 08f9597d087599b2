"""ctypes binding of libquorum_hip.so (include/quorum_hip.h). Fails loudly when the
library is missing: there is no fallback path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libquorum_hip.so")

QH_OK, QH_EBADARG, QH_EFULL, QH_EIO, QH_EHIP, QH_EFORMAT, QH_ENOMEM = range(7)
STATUS_NAMES = ["QH_OK", "QH_EBADARG", "QH_EFULL", "QH_EIO", "QH_EHIP", "QH_EFORMAT", "QH_ENOMEM"]


class QhError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("%s: %s" % (STATUS_NAMES[status] if 0 <= status < len(STATUS_NAMES) else status, msg))
        self.status = status


class QhRecord(C.Structure):
    _fields_ = [("seq_off", C.c_uint32), ("len", C.c_uint32), ("qual_off", C.c_uint32), ("reserved", C.c_uint32)]


class QhBuildStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("instances", "hq_instances", "distinct", "slots", "records", "grows", "foreign", "window_passes", "early_preps")]


class QhDbInfo(C.Structure):
    _fields_ = [("size", C.c_uint64), ("key_bytes", C.c_uint64), ("value_bytes", C.c_uint64), ("offset", C.c_uint64),
                ("key_len", C.c_uint32), ("val_len", C.c_uint32), ("max_reprobe", C.c_uint32), ("bits", C.c_uint32),
                ("shard_bits", C.c_uint32), ("window_bits", C.c_uint32), ("format", C.c_char * 32)]


class QhCorrectParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("skip", "good", "anchor", "min_count", "cutoff", "window", "error", "qual_cutoff",
                 "trim_contaminant", "do_homo_trim", "homo_trim", "no_discard")] + \
               [("collision_prob", C.c_double), ("poisson_threshold", C.c_double)]


class QhCorrectStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("reads", "corrected", "skipped", "substitutions", "truncations", "bases_in", "bases_out")]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "quorum_amd: %s is missing. Build it with `make -C quorum_amd/csrc` "
        "(or python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback." % LIB_PATH)

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

_p, _u64p, _u8p, _i32p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)

# every symbol include/quorum_hip.h declares: (name, restype, argtypes)
SIGNATURES = [
    ("qh_last_error", C.c_char_p, []),
    ("qh_version", C.c_char_p, []),
    ("qh_build_create", C.c_int, [C.POINTER(_p), C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("qh_build_destroy", None, [_p]),
    ("qh_build_set_stream", C.c_int, [_p, _p]),
    ("qh_build_add_fastq_device", C.c_int, [_p, _p, C.c_size_t]),
    ("qh_build_add_fastq", C.c_int, [_p, C.c_char_p, C.c_size_t]),
    ("qh_build_add_fastq_host_parsed", C.c_int, [_p, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_size_t)]),
    ("qh_build_add_fasta", C.c_int, [_p, C.c_char_p, C.c_size_t]),
    ("qh_build_add_records_device", C.c_int, [_p, _p, C.c_size_t, C.POINTER(QhRecord), C.c_size_t]),
    ("qh_build_add_mers", C.c_int, [_p, _u64p, _u8p, C.c_size_t]),
    ("qh_build_add_mers_device", C.c_int, [_p, _p, C.c_size_t]),
    ("qh_build_add_hashed_device", C.c_int, [_p, _p, C.c_size_t]),
    ("qh_build_add_mers_device_noalloc", C.c_int, [_p, _p, C.c_size_t]),
    ("qh_detect_min_qual_char", C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    ("qh_mer_hash", C.c_int, [C.c_int, _u64p, C.c_size_t, _u64p]),
    ("qh_host_alloc", C.c_int, [C.POINTER(_p), C.c_size_t]),
    ("qh_host_free", None, [_p]),
    ("qh_build_sync", C.c_int, [_p]),
    ("qh_build_finish", C.c_int, [_p, C.POINTER(QhBuildStats)]),
    ("qh_build_get_stats", C.c_int, [_p, C.POINTER(QhBuildStats)]),
    ("qh_build_reset", C.c_int, [_p]),
    ("qh_build_flush", C.c_int, [_p]),
    ("qh_build_write", C.c_int, [_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p)]),
    ("qh_build_write_shard", C.c_int, [_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.c_int]),
    ("qh_build_grow_to", C.c_int, [_p, C.c_int]),
    ("qh_build_write_jf", C.c_int, [_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p)]),
    ("qh_jf_write_file", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_int, _u64p, _u64p, C.c_uint64, C.c_int,
                                   C.POINTER(C.c_char_p)]),
    ("qh_jf_read_file", C.c_int, [C.c_char_p, _i32p, _i32p, _u64p, _u64p, _u64p, C.c_uint64, _u64p]),
    ("qh_fastq_4line_prefix", C.c_size_t, [C.c_char_p, C.c_size_t, C.c_int]),
    ("qh_fastq_split_records", C.c_int, [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_size_t)]),
    ("qh_build_dump", C.c_int, [_p, _u64p, _u64p, C.c_uint64, _u64p]),
    ("qh_build_as_query", C.c_int, [_p, C.POINTER(_p)]),
    ("qh_build_extract_partitioned_device", C.c_int, [_p, _p, C.c_size_t, _p, C.c_uint64, _p]),
    ("qh_multi_create", C.c_int, [C.POINTER(_p), C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, _i32p]),
    ("qh_multi_destroy", None, [_p]),
    ("qh_multi_add_fastq", C.c_int, [_p, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_size_t)]),
    ("qh_multi_finish", C.c_int, [_p, C.POINTER(QhBuildStats)]),
    ("qh_multi_write", C.c_int, [_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p)]),
    ("qh_multi_dump", C.c_int, [_p, _u64p, _u64p, C.c_uint64, _u64p]),
    ("qh_multi_info", C.c_int, [_p, C.POINTER(C.c_int), C.POINTER(C.c_int), _u64p, _u64p, _u64p]),
    ("qh_db_header_info", C.c_int, [C.c_char_p, C.POINTER(QhDbInfo)]),
    ("qh_db_header_parse", C.c_int, [_p, C.c_size_t, C.POINTER(QhDbInfo)]),
    ("qh_query_open", C.c_int, [C.POINTER(_p), C.c_char_p, C.c_int]),
    ("qh_query_close", None, [_p]),
    ("qh_query_info", C.c_int, [_p, C.POINTER(C.c_int), C.POINTER(C.c_int), _u64p]),
    ("qh_query_batch", C.c_int, [_p, _u64p, C.c_size_t, _u64p]),
    ("qh_query_batch_device", C.c_int, [_p, _p, C.c_size_t, _p, _p]),
    ("qh_query_get_val_batch", C.c_int, [_p, _u64p, C.c_size_t, _u64p]),
    ("qh_query_best_alternatives_batch", C.c_int, [_p, _u64p, _u8p, C.c_size_t, _u64p, _i32p, _i32p, _i32p]),
    ("qh_query_stats", C.c_int, [_p, _u64p, _u64p]),
    ("qh_query_histo", C.c_int, [_p, _u64p]),
    ("qh_poisson_term", C.c_double, [C.c_double, C.c_uint]),
    ("qh_query_poisson_cutoff", C.c_int, [_p, C.c_double, C.c_double, C.POINTER(C.c_uint)]),
    ("qh_contaminant_open_fasta", C.c_int, [C.POINTER(_p), C.c_char_p, C.c_int, C.c_int]),
    ("qh_query_is_member_batch", C.c_int, [_p, _u64p, C.c_size_t, _u8p]),
    ("qh_query_batch_contaminant", C.c_int, [_p, _p, _u64p, C.c_size_t, _u64p, _u8p]),
    ("qh_query_batch_contaminant_device", C.c_int, [_p, _p, _p, C.c_size_t, _p, _p, _p]),
    ("qh_query_dump", C.c_int, [_p, _u64p, _u64p, C.c_uint64, _u64p]),
    ("qh_corrector_create", C.c_int, [C.POINTER(_p), _p, _p, C.POINTER(QhCorrectParams)]),
    ("qh_corrector_destroy", None, [_p]),
    ("qh_corrector_correct_device", C.c_int, [_p, _p, C.c_size_t, _p]),
    ("qh_corrector_fetch", C.c_int, [_p, C.POINTER(_p), C.POINTER(C.c_size_t), C.POINTER(_p), C.POINTER(C.c_size_t),
                                     C.POINTER(QhCorrectStats)]),
    ("qh_corrector_correct_fastq", C.c_int, [_p, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_size_t)]),
    ("qh_corrector_kernel_time", C.c_int, [_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("qh_build_enable_timing", C.c_int, [_p, C.c_int]),
    ("qh_build_kernel_time", C.c_int, [_p, C.c_int, C.POINTER(C.c_double), _u64p]),
    ("qh_build_set_mode", C.c_int, [_p, C.c_int, C.c_int]),
    ("qh_build_set_defer", C.c_int, [_p, C.c_longlong]),
    ("qh_bench_lds_rate", C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    ("qh_bench_random_access", C.c_int, [C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_double)]),
]

for _name, _res, _args in SIGNATURES:
    _f = getattr(lib, _name)  # AttributeError here = the .so does not export a declared symbol
    _f.restype = _res
    _f.argtypes = _args


def last_error():
    return lib.qh_last_error().decode("utf-8", "replace")


def check(status):
    if status != QH_OK:
        raise QhError(status, last_error())
