import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quorum_amd._lib import lib, check
names = ["ds_read_b64", "ds_write_b64", "ds_cmpst_rtn_b64", "ds_cmpst_rtn_b32", "ds_add_rtn_u32", "ds_add_u32", "ds_add_rtn_u64", "read+write+read b64"]
for m, n in enumerate(names):
    r = C.c_double()
    check(lib.qh_bench_lds_rate(0, m, 2000, C.byref(r)))
    print("%-22s %8.1f G lane-ops/s  = %.2f lanes/clk/CU @2.4GHz" % (n, r.value / 1e9, r.value / 256 / 2.4e9), flush=True)
