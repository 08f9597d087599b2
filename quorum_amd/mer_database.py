"""Python mirror of the reference's ``hash_with_quality`` and ``database_query``
(src/mer_database.hpp:65-188, 251-362) over the libquorum_hip C ABI.

Same names and argument meaning as upstream where Python allows: ``add(mer, quality)``,
``done()``, ``write(path)``; ``db[mer] -> (count, hq)``, ``get_val``,
``get_best_alternatives``, iteration over ``(mer, (count, hq))``.  Mers are the
2-bit packed integers of jellyfish::mer_dna (A=0 C=1 G=2 T=3, last character in the
two low bits); ``mer_from_str`` / ``mer_to_str`` convert.
"""
import ctypes as C
import numpy as np

from . import _lib
from ._lib import lib, check

_CODE = {"A": 0, "C": 1, "G": 2, "T": 3, "a": 0, "c": 1, "g": 2, "t": 3}


def mer_from_str(s):
    m = 0
    for ch in s:
        m = (m << 2) | _CODE[ch]
    return m


def mer_to_str(m, k):
    return "".join("ACGT"[(m >> (2 * (k - 1 - i))) & 3] for i in range(k))


def revcomp(m, k):
    r = 0
    for _ in range(k):
        r = (r << 2) | (3 - (m & 3))
        m >>= 2
    return r


def canonical(m, k):
    r = revcomp(m, k)
    return m if m < r else r


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint64))


class HashWithQuality:
    """hash_with_quality(size, key_len, bits, nb_threads, reprobe_limit) (mer_database.hpp:78).

    key_len is 2*k bits as upstream; nb_threads is accepted and ignored.  min_qual is
    only needed for the FASTQ entry points (quality_mer_counter's qual_thresh,
    create_database.cc:50-54)."""

    def __init__(self, size, key_len, bits, nb_threads=1, reprobe_limit=126, min_qual=0, device=0,
                 shard_bits=0, shard_id=0):
        if key_len % 2:
            raise ValueError("key_len must be 2*k")
        self.k = key_len // 2
        self.bits = bits
        self._h = C.c_void_p()
        check(lib.qh_build_create(C.byref(self._h), self.k, bits, size, min_qual, reprobe_limit, device,
                                  shard_bits, shard_id))

    # -- upstream surface -------------------------------------------------------------
    def add(self, mer, quality):
        """bool add(const mer_dna&, unsigned quality) (mer_database.hpp:94). Accepts
        scalars or arrays; returns False when the table is full and cannot grow."""
        mers = np.atleast_1d(np.asarray(mer, dtype=np.uint64))
        q = np.ascontiguousarray(np.broadcast_to(np.asarray(quality, dtype=np.uint8), mers.shape))
        mers, pm = _u64(mers)
        st = lib.qh_build_add_mers(self._h, pm, q.ctypes.data_as(C.POINTER(C.c_uint8)), mers.size)
        if st == _lib.QH_OK:
            st = lib.qh_build_sync(self._h)
        if st == _lib.QH_EFULL:
            return False
        check(st)
        return True

    def done(self):
        """void done() (mer_database.hpp:128): end of input; returns the statistics."""
        st = _lib.QhBuildStats()
        check(lib.qh_build_finish(self._h, C.byref(st)))
        return {n: getattr(st, n) for n, _ in st._fields_}

    def write(self, path, argv=()):
        """void write(std::ostream&, database_header*) (mer_database.hpp:115)."""
        arr = (C.c_char_p * max(1, len(argv)))(*[a.encode() for a in argv])
        check(lib.qh_build_write(self._h, str(path).encode(), len(argv), arr))

    def write_jf(self, path, argv=()):
        """The same database with the keys image in the recalled jellyfish::large_hash::array layout
        (qh_jflayout.h; PARITY UNPINNED, not checked against any Jellyfish reader or writer)."""
        arr = (C.c_char_p * max(1, len(argv)))(*[a.encode() for a in argv])
        check(lib.qh_build_write_jf(self._h, str(path).encode(), len(argv), arr))

    # -- chunk entry points (quality_mer_counter::start, create_database.cc:56-95) -----
    def add_fastq(self, data):
        data = bytes(data)
        check(lib.qh_build_add_fastq(self._h, data, len(data)))

    def add_fastq_device(self, dev_ptr, nbytes):
        check(lib.qh_build_add_fastq_device(self._h, C.c_void_p(dev_ptr), nbytes))

    def add_fastq_host_parsed(self, data, final=True):
        data = bytes(data)
        used = C.c_size_t()
        check(lib.qh_build_add_fastq_host_parsed(self._h, data, len(data), 1 if final else 0, C.byref(used)))
        return used.value

    def add_fasta(self, data):
        """every canonical k-mer of a FASTA buffer, quality bit set (contaminant set input,
        error_correct_reads.cc:86-90)"""
        data = bytes(data)
        check(lib.qh_build_add_fasta(self._h, data, len(data)))

    def add_mers_device(self, dev_ptr, n):
        check(lib.qh_build_add_mers_device(self._h, C.c_void_p(dev_ptr), n))

    def add_mers_device_noalloc(self, dev_ptr, n):
        """qh_build_add_mers_device without any allocation (counters and overflow queue in the library's
        file-scope device array; at most 2^24 items; the table cannot grow)"""
        check(lib.qh_build_add_mers_device_noalloc(self._h, C.c_void_p(dev_ptr), n))

    def add_hashed_device(self, dev_ptr, n):
        """items in the library's hashed exchange form (qh_build_extract_partitioned_device)"""
        check(lib.qh_build_add_hashed_device(self._h, C.c_void_p(dev_ptr), n))

    def set_stream(self, stream_handle):
        check(lib.qh_build_set_stream(self._h, C.c_void_p(stream_handle)))

    def sync(self):
        check(lib.qh_build_sync(self._h))

    def enable_timing(self, on=True):
        check(lib.qh_build_enable_timing(self._h, 1 if on else 0))

    KERNELS = ("k_build_direct/k_insert_items", "k_partition<FASTQ|ITEMS> level 1", "k_partition<ITEMS> level 2",
               "k_window_build")

    def kernel_time(self, which=0):
        """(total ms, launches) of one of the main kernels since enable_timing():
        0 direct build / insert, 1 partition level 1, 2 partition level 2, 3 window build"""
        ms, n = C.c_double(), C.c_uint64()
        check(lib.qh_build_kernel_time(self._h, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def set_mode(self, mode="auto", level1_bits=-1):
        check(lib.qh_build_set_mode(self._h, {"auto": 0, "direct": 1, "partition": 2}[mode], level1_bits))

    def set_defer(self, max_bytes=-1):
        """memory the window buckets of a batch of partitioned chunks may take before the table is streamed
        (0: every chunk at once, negative: automatic)"""
        check(lib.qh_build_set_defer(self._h, max_bytes))

    def reset(self):
        check(lib.qh_build_reset(self._h))

    def flush(self):
        """starts the pass over the table for what waits in the window buckets, without waiting for it"""
        check(lib.qh_build_flush(self._h))

    def stats(self):
        st = _lib.QhBuildStats()
        check(lib.qh_build_get_stats(self._h, C.byref(st)))
        return {n: getattr(st, n) for n, _ in st._fields_}

    def dump(self):
        """Every (mer, count<<1|hq) of the table as two uint64 arrays (unordered)."""
        n = C.c_uint64()
        check(lib.qh_build_dump(self._h, None, None, 0, C.byref(n)))
        mers = np.empty(n.value, dtype=np.uint64)
        vals = np.empty(n.value, dtype=np.uint64)
        if n.value:
            check(lib.qh_build_dump(self._h, mers.ctypes.data_as(C.POINTER(C.c_uint64)),
                                    vals.ctypes.data_as(C.POINTER(C.c_uint64)), n.value, C.byref(n)))
        return mers, vals

    def as_query(self):
        q = C.c_void_p()
        check(lib.qh_build_as_query(self._h, C.byref(q)))
        return DatabaseQuery(_handle=q, _owner=self)

    def close(self):
        if self._h:
            lib.qh_build_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def detect_min_qual_char(data):
    """the quorum driver's quality-offset detection over the first 1000 records (quorum.in:129-152);
    raises QhError for an unusual minimum like the driver stops"""
    data = bytes(data)
    c = C.c_int()
    check(lib.qh_detect_min_qual_char(data, len(data), C.byref(c)))
    return c.value


def jf_write_file(path, k, bits, size, mers, vals, reprobe_limit=126, argv=()):
    """host-only: n distinct (mer, count<<1|hq) pairs -> a database file whose keys image follows the recalled
    jellyfish::large_hash::array layout (qh_jflayout.h; PARITY UNPINNED)"""
    mers, pm = _u64(mers)
    vals, pv = _u64(vals)
    arr = (C.c_char_p * max(1, len(argv)))(*[a.encode() for a in argv])
    check(lib.qh_jf_write_file(str(path).encode(), k, bits, size, reprobe_limit, pm, pv, mers.size, len(argv), arr))


def jf_read_file(path):
    """host-only: (k, bits, size, mers, vals) of a file written by jf_write_file / HashWithQuality.write_jf"""
    k, bits, size, n = C.c_int32(), C.c_int32(), C.c_uint64(), C.c_uint64()
    check(lib.qh_jf_read_file(str(path).encode(), C.byref(k), C.byref(bits), C.byref(size), None, None, 0, C.byref(n)))
    mers = np.empty(n.value, dtype=np.uint64)
    vals = np.empty(n.value, dtype=np.uint64)
    if n.value:
        check(lib.qh_jf_read_file(str(path).encode(), C.byref(k), C.byref(bits), C.byref(size),
                                  mers.ctypes.data_as(C.POINTER(C.c_uint64)),
                                  vals.ctypes.data_as(C.POINTER(C.c_uint64)), n.value, C.byref(n)))
    return k.value, bits.value, size.value, mers, vals


def poisson_term(lam, i):
    """double poisson_term(double lambda, unsigned int i) (error_correct_reads.cc:53-61)"""
    return lib.qh_poisson_term(lam, i)


def contaminant_database(fasta_path, k, device=0):
    """contaminant_database (error_correct_reads.cc:83-99,692-708) built from FASTA at mer
    length k; use .is_member() or DatabaseQuery.lookup_with_contaminant()."""
    q = C.c_void_p()
    st = lib.qh_contaminant_open_fasta(C.byref(q), str(fasta_path).encode(), k, device)
    if st != _lib.QH_OK:
        raise RuntimeError(_lib.last_error())
    return DatabaseQuery(_handle=q)


class DatabaseQuery:
    """database_query(filename, no_mmap=false) (mer_database.hpp:270)."""

    def __init__(self, filename=None, no_mmap=False, device=0, _handle=None, _owner=None):
        self._owner = _owner  # keeps a live build handle alive
        if _handle is not None:
            self._q = _handle
        else:
            self._q = C.c_void_p()
            st = lib.qh_query_open(C.byref(self._q), str(filename).encode(), device)
            if st != _lib.QH_OK:
                # upstream throws std::runtime_error on open/parse failure (:259-263)
                raise RuntimeError(_lib.last_error())
        k, bits, size = C.c_int(), C.c_int(), C.c_uint64()
        check(lib.qh_query_info(self._q, C.byref(k), C.byref(bits), C.byref(size)))
        self.k, self.bits, self.size = k.value, bits.value, size.value

    def header(self):
        return {"bits": self.bits, "key_len": 2 * self.k, "size": self.size}

    def lookup(self, mers):
        """operator[] batched: count<<1|hq per mer (0 = absent)."""
        mers, pm = _u64(np.atleast_1d(mers))
        out = np.empty(mers.size, dtype=np.uint64)
        check(lib.qh_query_batch(self._q, pm, mers.size, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def __getitem__(self, mer):
        """std::pair<uint64_t,int> operator[](const mer_dna&) (mer_database.hpp:284)."""
        v = int(self.lookup([mer])[0])
        return (v >> 1, v & 1)

    def get_val(self, mers):
        """uint64_t get_val(const mer_dna&) (mer_database.hpp:296), scalar or batched."""
        scalar = np.isscalar(mers)
        m, pm = _u64(np.atleast_1d(mers))
        out = np.empty(m.size, dtype=np.uint64)
        check(lib.qh_query_get_val_batch(self._q, pm, m.size, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return int(out[0]) if scalar else out

    def get_best_alternatives(self, fmers, backward=0):
        """int get_best_alternatives(mer_type&, uint64_t counts[4], int& ucode, int& level)
        (mer_database.hpp:303) batched; returns (nkept, counts[n,4], ucode, level)."""
        m, pm = _u64(np.atleast_1d(fmers))
        bw = np.ascontiguousarray(np.broadcast_to(np.asarray(backward, dtype=np.uint8), m.shape))
        counts = np.empty((m.size, 4), dtype=np.uint64)
        ucode = np.empty(m.size, dtype=np.int32)
        level = np.empty(m.size, dtype=np.int32)
        nkept = np.empty(m.size, dtype=np.int32)
        i32 = C.POINTER(C.c_int32)
        check(lib.qh_query_best_alternatives_batch(
            self._q, pm, bw.ctypes.data_as(C.POINTER(C.c_uint8)), m.size,
            counts.ctypes.data_as(C.POINTER(C.c_uint64)), ucode.ctypes.data_as(i32),
            level.ctypes.data_as(i32), nkept.ctypes.data_as(i32)))
        return nkept, counts, ucode, level

    def stats(self):
        d, t = C.c_uint64(), C.c_uint64()
        check(lib.qh_query_stats(self._q, C.byref(d), C.byref(t)))
        return d.value, t.value

    def poisson_cutoff(self, collision_prob, poisson_threshold):
        """compute_poisson_cutoff(vals(), collision_prob, poisson_threshold)
        (error_correct_reads.cc:650-674)"""
        c = C.c_uint()
        check(lib.qh_query_poisson_cutoff(self._q, collision_prob, poisson_threshold, C.byref(c)))
        return c.value

    def is_member(self, mers):
        """contaminant_database::is_contaminant (error_correct_reads.cc:94), batched"""
        m, pm = _u64(np.atleast_1d(mers))
        flags = np.empty(m.size, dtype=np.uint8)
        check(lib.qh_query_is_member_batch(self._q, pm, m.size, flags.ctypes.data_as(C.POINTER(C.c_uint8))))
        return flags.astype(bool)

    def lookup_with_contaminant(self, contaminant, mers):
        """database lookup + is_contaminant of the same canonical mers in one pass
        (error_correct_reads.cc:348,390,620) -> (values, flags)"""
        m, pm = _u64(np.atleast_1d(mers))
        out = np.empty(m.size, dtype=np.uint64)
        flags = np.empty(m.size, dtype=np.uint8)
        check(lib.qh_query_batch_contaminant(self._q, contaminant._q, pm, m.size,
                                             out.ctypes.data_as(C.POINTER(C.c_uint64)),
                                             flags.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out, flags.astype(bool)

    def histo(self):
        h = np.zeros((1001, 2), dtype=np.uint64)
        check(lib.qh_query_histo(self._q, h.ctypes.data_as(C.POINTER(C.c_uint64))))
        return h

    def dump(self):
        n = C.c_uint64()
        check(lib.qh_query_dump(self._q, None, None, 0, C.byref(n)))
        mers = np.empty(n.value, dtype=np.uint64)
        vals = np.empty(n.value, dtype=np.uint64)
        if n.value:
            check(lib.qh_query_dump(self._q, mers.ctypes.data_as(C.POINTER(C.c_uint64)),
                                    vals.ctypes.data_as(C.POINTER(C.c_uint64)), n.value, C.byref(n)))
        return mers, vals

    def __iter__(self):
        """begin()/end() (mer_database.hpp:360-361): yields (mer, (count, hq))."""
        mers, vals = self.dump()
        for m, v in zip(mers.tolist(), vals.tolist()):
            yield m, (v >> 1, v & 1)

    def close(self):
        if self._q:
            lib.qh_query_close(self._q)
            self._q = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
