"""ctypes wrapper of the CPU oracle (oracle/quorum_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg. Nothing under quorum_amd/ may import this module."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libquorum_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


if not os.path.exists(_SO):
    build()
_o = C.CDLL(_SO)

_u64p, _u8p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)


class QoRec(C.Structure):
    _fields_ = [("seq_off", C.c_uint64), ("qual_off", C.c_uint64), ("len", C.c_uint32), ("in_scratch", C.c_uint32)]


_o.qo_code.restype = C.c_int
_o.qo_code.argtypes = [C.c_int]
_o.qo_revcomp.restype = C.c_uint64
_o.qo_revcomp.argtypes = [C.c_uint64, C.c_int]
_o.qo_canonical.restype = C.c_uint64
_o.qo_canonical.argtypes = [C.c_uint64, C.c_int]
_o.qo_extract_read.restype = C.c_size_t
_o.qo_extract_read.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int, _u64p, _u64p, _u64p, _u8p]
_o.qo_parse_fastq.restype = C.c_int64
_o.qo_parse_fastq.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(QoRec), C.c_size_t, C.c_char_p, C.c_size_t]
_o.qo_db_create.restype = C.c_void_p
_o.qo_db_create.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int]
_o.qo_db_destroy.argtypes = [C.c_void_p]
_o.qo_db_add.restype = C.c_int
_o.qo_db_add.argtypes = [C.c_void_p, C.c_uint64, C.c_uint]
_o.qo_db_add_mers.restype = C.c_int64
_o.qo_db_add_mers.argtypes = [C.c_void_p, _u64p, _u8p, C.c_size_t]
_o.qo_db_add_fastq.restype = C.c_int64
_o.qo_db_add_fastq.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int]
_o.qo_db_add_fastq_mt.restype = C.c_int64
_o.qo_db_add_fastq_mt.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int]
_o.qo_db_size.restype = C.c_uint64
_o.qo_db_size.argtypes = [C.c_void_p]
_o.qo_db_distinct.restype = C.c_uint64
_o.qo_db_distinct.argtypes = [C.c_void_p]
_o.qo_db_lookup.argtypes = [C.c_void_p, _u64p, C.c_size_t, _u64p]
_o.qo_db_get_val.restype = C.c_uint64
_o.qo_db_get_val.argtypes = [C.c_void_p, C.c_uint64]
_o.qo_db_best_alternatives.restype = C.c_int
_o.qo_db_best_alternatives.argtypes = [C.c_void_p, C.c_uint64, C.c_int, _u64p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
_o.qo_db_dump.restype = C.c_uint64
_o.qo_db_dump.argtypes = [C.c_void_p, _u64p, _u64p, C.c_uint64]
_o.qo_db_stats.argtypes = [C.c_void_p, _u64p, _u64p]
_o.qo_db_histo.argtypes = [C.c_void_p, _u64p]
_o.qo_poisson_term.restype = C.c_double
_o.qo_poisson_term.argtypes = [C.c_double, C.c_uint]
_o.qo_poisson_cutoff_from.restype = C.c_uint
_o.qo_poisson_cutoff_from.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_double]
_o.qo_db_poisson_cutoff.restype = C.c_uint
_o.qo_db_poisson_cutoff.argtypes = [C.c_void_p, C.c_double, C.c_double]
_o.qo_db_add_fasta.restype = C.c_int64
_o.qo_db_add_fasta.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]



class QoEcParams(C.Structure):
    """qo_ec_params of oracle/error_correct_oracle.c (switches of error_correct_reads_cmdline.yaggo)"""
    _fields_ = [("skip", C.c_int), ("good", C.c_int), ("anchor", C.c_int), ("min_count", C.c_int), ("cutoff", C.c_int),
                ("window", C.c_int), ("error", C.c_int), ("qual_cutoff", C.c_int), ("trim_contaminant", C.c_int),
                ("do_homo_trim", C.c_int), ("homo_trim", C.c_int), ("no_discard", C.c_int),
                ("collision_prob", C.c_double), ("poisson_threshold", C.c_double)]


def ec_params(cutoff, skip=1, good=2, anchor=3, min_count=1, window=10, error=3, qual_cutoff=127,
              trim_contaminant=False, homo_trim=None, no_discard=False, apriori_error_rate=0.01,
              poisson_threshold=1e-6):
    """defaults of error_correct_reads_cmdline.yaggo:9-62 and error_correct_reads.cc:684-685,734-735"""
    return QoEcParams(skip, good, anchor, min_count, cutoff, window, error, qual_cutoff, int(trim_contaminant),
                      int(homo_trim is not None), 0 if homo_trim is None else homo_trim, int(no_discard),
                      apriori_error_rate / 3, poisson_threshold)


_o.qo_ec_correct_fastq.restype = C.c_int64
_o.qo_ec_correct_fastq.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(QoEcParams), C.c_char_p, C.c_size_t,
                                   C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t),
                                   C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
_o.qo_ec_print_log.restype = C.c_size_t
_o.qo_ec_print_log.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_size_t, C.c_char_p]


def ec_print_log(direction, entries):
    """err_log's operator<< (err_log.hpp:111-135); entries = [(type, pos, from, to)], type 0 sub / 1 trunc"""
    flat = []
    for t, pos, f, to in entries:
        flat += [t, pos, ord(f) if f else 0, ord(to) if to else 0]
    arr = (C.c_int * max(1, len(flat)))(*flat)
    out = C.create_string_buffer(32 * len(entries) + 16)
    n = _o.qo_ec_print_log(direction, arr, len(entries), out)
    return out.raw[:n]


PARSE_ERRORS = {-1: "Truncated fastq file", -2: "Invalid fastq file: wrong number of quals",
                -3: "Invalid fastq file: header missing", -4: "Unsupported format", -5: "out of capacity",
                -10: "Hash is full"}


def revcomp(m, k):
    return _o.qo_revcomp(m, k)


def canonical(m, k):
    return _o.qo_canonical(m, k)


def poisson_term(lam, i):
    """error_correct_reads.cc:53-61"""
    return _o.qo_poisson_term(lam, i)


def poisson_cutoff_from(distinct, total, collision_prob, poisson_threshold):
    """error_correct_reads.cc:660-667 from the (distinct, total) reduction"""
    return _o.qo_poisson_cutoff_from(distinct, total, collision_prob, poisson_threshold)


def extract_read(seq, qual, k, thresh):
    """create_database.cc:64-92 on one read -> (mers uint64[], hq uint8[])."""
    seq, qual = bytes(seq), bytes(qual)
    n = len(seq)
    mers = np.empty(n + 1, dtype=np.uint64)
    hq = np.empty(n + 1, dtype=np.uint8)
    m, rm = C.c_uint64(0), C.c_uint64(0)
    cnt = _o.qo_extract_read(seq, qual, n, k, thresh, C.byref(m), C.byref(rm),
                             mers.ctypes.data_as(_u64p), hq.ctypes.data_as(_u8p))
    return mers[:cnt].copy(), hq[:cnt].copy()


def parse_fastq(data):
    """-> list of (seq bytes, qual bytes); raises ValueError with upstream's message."""
    data = bytes(data)
    cap = len(data) // 4 + 2
    recs = (QoRec * cap)()
    scratch = C.create_string_buffer(len(data) + 1)
    n = _o.qo_parse_fastq(data, len(data), recs, cap, scratch, len(data) + 1)
    if n < 0:
        raise ValueError(PARSE_ERRORS.get(n, str(n)))
    out = []
    for i in range(n):
        r = recs[i]
        base = scratch.raw if r.in_scratch else data
        out.append((base[r.seq_off:r.seq_off + r.len], base[r.qual_off:r.qual_off + r.len]))
    return out


class OracleDB:
    """hash_with_quality + database_query restated on the CPU."""

    def __init__(self, size, k, bits, reprobe_limit=126):
        self.k, self.bits = k, bits
        self._db = _o.qo_db_create(size, k, bits, reprobe_limit)
        if not self._db:
            raise ValueError("bad oracle geometry")

    def add(self, mer, quality):
        return _o.qo_db_add(self._db, int(mer), int(quality)) == 0

    def add_mers(self, mers, hq):
        mers = np.ascontiguousarray(mers, dtype=np.uint64)
        hq = np.ascontiguousarray(hq, dtype=np.uint8)
        r = _o.qo_db_add_mers(self._db, mers.ctypes.data_as(_u64p), hq.ctypes.data_as(_u8p), mers.size)
        if r < 0:
            raise RuntimeError("Hash is full")

    def add_fastq(self, data, thresh, threads=0):
        data = bytes(data)
        if threads and threads > 0:
            r = _o.qo_db_add_fastq_mt(self._db, data, len(data), thresh, threads)
        else:
            r = _o.qo_db_add_fastq(self._db, data, len(data), thresh)
        if r < 0:
            raise ValueError(PARSE_ERRORS.get(r, str(r)))
        return r

    def add_fastq_ptr(self, ptr, nbytes, thresh, threads):
        r = _o.qo_db_add_fastq_mt(self._db, C.cast(ptr, C.c_char_p), nbytes, thresh, threads)
        if r < 0:
            raise ValueError(PARSE_ERRORS.get(r, str(r)))
        return r

    def add_fasta(self, data):
        """contaminant set: every canonical k-mer of a FASTA file, stored with quality 1"""
        data = bytes(data)
        r = _o.qo_db_add_fasta(self._db, data, len(data))
        if r < 0:
            raise ValueError(PARSE_ERRORS.get(r, str(r)))
        return r

    def poisson_cutoff(self, collision_prob, poisson_threshold):
        return _o.qo_db_poisson_cutoff(self._db, collision_prob, poisson_threshold)

    def size(self):
        return _o.qo_db_size(self._db)

    def distinct(self):
        return _o.qo_db_distinct(self._db)

    def lookup(self, mers):
        mers = np.ascontiguousarray(np.atleast_1d(mers), dtype=np.uint64)
        out = np.empty(mers.size, dtype=np.uint64)
        _o.qo_db_lookup(self._db, mers.ctypes.data_as(_u64p), mers.size, out.ctypes.data_as(_u64p))
        return out

    def get_val(self, mer):
        return _o.qo_db_get_val(self._db, int(mer))

    def best_alternatives(self, fmer, backward=0):
        counts = (C.c_uint64 * 4)()
        ucode, level = C.c_int(0), C.c_int(0)
        n = _o.qo_db_best_alternatives(self._db, int(fmer), int(backward), counts, C.byref(ucode), C.byref(level))
        return n, list(counts), ucode.value, level.value

    def dump(self):
        n = self.distinct()
        mers = np.empty(n, dtype=np.uint64)
        vals = np.empty(n, dtype=np.uint64)
        got = _o.qo_db_dump(self._db, mers.ctypes.data_as(_u64p), vals.ctypes.data_as(_u64p), n)
        return mers[:got], vals[:got]

    def sorted_dump(self):
        m, v = self.dump()
        o = np.argsort(m, kind="stable")
        return m[o], v[o]

    def stats(self):
        d, t = C.c_uint64(), C.c_uint64()
        _o.qo_db_stats(self._db, C.byref(d), C.byref(t))
        return d.value, t.value

    def histo(self):
        h = np.zeros((1001, 2), dtype=np.uint64)
        _o.qo_db_histo(self._db, h.ctypes.data_as(_u64p))
        return h

    def correct_fastq(self, data, params, contaminant=None):
        """quorum_error_correct_reads (one thread) over a FASTQ buffer -> (fa bytes, log bytes, reads)"""
        data = bytes(data)
        cap = 3 * len(data) + 4096
        fa, lg = C.create_string_buffer(cap), C.create_string_buffer(cap)
        nfa, nlg = C.c_size_t(), C.c_size_t()
        n = _o.qo_ec_correct_fastq(self._db, contaminant._db if contaminant is not None else None, C.byref(params),
                                   data, len(data), fa, cap, C.byref(nfa), lg, cap, C.byref(nlg))
        if n < 0:
            raise ValueError(PARSE_ERRORS.get(n, str(n)))
        return fa.raw[:nfa.value], lg.raw[:nlg.value], n

    def close(self):
        if self._db:
            _o.qo_db_destroy(self._db)
            self._db = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
