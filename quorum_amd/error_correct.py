"""Python mirror of the reference's read corrector (``error_correct_t`` configured as in
``main`` of src/error_correct_reads.cc:676-742) over the libquorum_hip C ABI.

Keyword names are the long switch names of src/error_correct_reads_cmdline.yaggo with
``-`` replaced by ``_``; defaults are the yaggo defaults.  ``cutoff=None`` computes the
Poisson cutoff from the database like upstream does when ``-p`` is not given (:712-718).
"""
import ctypes as C

from . import _lib
from ._lib import lib, check


class ErrorCorrector:
    def __init__(self, db, contaminant=None, min_count=1, skip=1, good=2, anchor_count=3, window=10, error=3,
                 trim_contaminant=False, homo_trim=None, apriori_error_rate=0.01, poisson_threshold=1e-6,
                 cutoff=None, qual_cutoff_value=None, qual_cutoff_char=None, no_discard=False):
        if qual_cutoff_char is not None and qual_cutoff_value is not None:
            raise ValueError("qual-cutoff-char conflicts with qual-cutoff-value")
        if qual_cutoff_char is not None and len(qual_cutoff_char) != 1:
            raise ValueError("The qual-cutoff-char must be one ASCII character.")  # :680-681
        if qual_cutoff_value is not None and qual_cutoff_value > 127:
            raise ValueError("The qual-cutoff-value must be in the range 0-127.")  # :682-683
        qc = ord(qual_cutoff_char) if qual_cutoff_char is not None else \
            (qual_cutoff_value if qual_cutoff_value is not None else 127)          # :684-685
        if cutoff is None:
            cutoff = db.poisson_cutoff(apriori_error_rate / 3, poisson_threshold / apriori_error_rate)  # :712-715
            if cutoff == 0:
                raise RuntimeError("Cutoff computation failed. Pass it explicitly with -p switch.")     # :717-718
        self.cutoff = cutoff
        self._db, self._contaminant = db, contaminant  # keep the handles alive
        self.params = _lib.QhCorrectParams(skip, good, anchor_count, min_count, cutoff, window, error, qc,
                                           int(trim_contaminant), int(homo_trim is not None),
                                           0 if homo_trim is None else homo_trim, int(no_discard),
                                           apriori_error_rate / 3, poisson_threshold)
        self._c = C.c_void_p()
        check(lib.qh_corrector_create(C.byref(self._c), db._q, contaminant._q if contaminant is not None else None,
                                      C.byref(self.params)))

    def _fetch(self):
        fa, lg = C.c_void_p(), C.c_void_p()
        nfa, nlg = C.c_size_t(), C.c_size_t()
        st = _lib.QhCorrectStats()
        check(lib.qh_corrector_fetch(self._c, C.byref(fa), C.byref(nfa), C.byref(lg), C.byref(nlg), C.byref(st)))
        out = C.string_at(fa, nfa.value) if nfa.value else b""
        log = C.string_at(lg, nlg.value) if nlg.value else b""
        return out, log, {n: getattr(st, n) for n, _ in st._fields_}

    def correct_fastq(self, data, final=True):
        """one chunk of FASTQ text -> (fa bytes, log bytes, stats, bytes consumed)"""
        data = bytes(data)
        used = C.c_size_t()
        check(lib.qh_corrector_correct_fastq(self._c, data, len(data), 1 if final else 0, C.byref(used)))
        fa, log, st = self._fetch()
        return fa, log, st, used.value

    def correct_device(self, dev_ptr, nbytes, stream=None):
        """enqueue the correction of a chunk resident in HBM (strict 4-line FASTQ)"""
        check(lib.qh_corrector_correct_device(self._c, C.c_void_p(dev_ptr), nbytes,
                                              C.c_void_p(stream) if stream else None))

    def fetch(self):
        """-> (fa bytes, log bytes, stats) of the chunk enqueued last"""
        return self._fetch()

    def kernel_time(self):
        """(k_correct_reads ms, whole chunk ms) of the last chunk, by HIP events on the corrector's stream"""
        a, b = C.c_double(), C.c_double()
        check(lib.qh_corrector_kernel_time(self._c, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if self._c:
            lib.qh_corrector_destroy(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
