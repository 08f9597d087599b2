"""Throughput of the batched database_query lookups (qh_query_batch_device, keys and results
resident in HBM) on the bench database, against the measured random 8-byte load rate."""
import ctypes as C, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quorum_amd as qa
from quorum_amd import synth, _lib

k, bits = 24, 7
g = synth.make_genome(5_000_000)
fq = synth.make_fastq(g, 2_000_000, 100)
chunk = torch.from_numpy(fq).to("cuda:0")
h = qa.HashWithQuality(50_000_000, 2 * k, bits, min_qual=38)
h.add_fastq_device(chunk.data_ptr(), chunk.numel())
st = h.done()
q = h.as_query()
mers, vals = h.dump()
rng = np.random.default_rng(1)
n = 100_000_000
hit = mers[rng.integers(0, mers.size, n // 2)]
miss = rng.integers(0, 4 ** k, n - n // 2, dtype=np.uint64)
keys = np.concatenate([hit, miss]); rng.shuffle(keys)
d_keys = torch.from_numpy(keys.view(np.int64)).to("cuda:0")
d_out = torch.empty(n, dtype=torch.int64, device="cuda:0")
lib = _lib.lib
def run():
    _lib.check(lib.qh_query_batch_device(q._q, C.c_void_p(d_keys.data_ptr()), n, C.c_void_p(d_out.data_ptr()), None))
    # the query handle's own stream: synchronise through a host read of one word
    torch.cuda.synchronize()
run()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(); ts.append(time.perf_counter() - t0)
out = d_out.cpu().numpy().view(np.uint64)
found = int((out != 0).sum())
ms = C.c_double()
_lib.check(lib.qh_bench_random_access(0, 26, 1 << 27, 1, 3, C.byref(ms)))
print(json.dumps({"lookups": n, "found": found, "best_s": min(ts), "lookups_per_s": n / min(ts),
                  "random_load8_per_s": (1 << 27) / (ms.value * 1e-3), "distinct": int(st["distinct"])}))
