"""quorum_amd -- MI355X-native k-mer database build / lookup for Quorum.

The compute path is the C-ABI shared library ``quorum_amd/csrc/libquorum_hip.so``
(hand-written gfx950 HIP kernels, include/quorum_hip.h).  This package is the thin
Python mirror of the reference's ``hash_with_quality`` / ``database_query`` classes
(src/mer_database.hpp) over that ABI, used by the tests, bench.py and the
multi-GPU driver.  There is no CPU fallback: importing ``quorum_amd._lib`` raises if
the library has not been built.
"""
from ._lib import lib, QhError, last_error  # noqa: F401
from .mer_database import (HashWithQuality, DatabaseQuery, mer_from_str, mer_to_str, revcomp, canonical,  # noqa: F401
                           poisson_term, contaminant_database, detect_min_qual_char, jf_write_file, jf_read_file)
from .error_correct import ErrorCorrector  # noqa: F401
