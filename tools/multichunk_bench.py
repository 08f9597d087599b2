"""Cost of a chunk into a table that already holds entries (the window pass loads, converts and stores every window)
against the same chunk into an empty table: BASELINE configs[1] geometry, two different chunks of 2 M reads."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quorum_amd as qa
from quorum_amd import synth

g = synth.make_genome(5_000_000)
chunks = [torch.from_numpy(synth.make_fastq(g, 2_000_000, 100, seed=42 + i, first_id=i * 2_000_000)).cuda() for i in range(2)]
torch.cuda.synchronize()
h = qa.HashWithQuality(50_000_000, 48, 7, min_qual=38)
rows = []
for rep in range(4):
    h.reset()
    row = {}
    for i, c in enumerate(chunks):
        h.enable_timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h.add_fastq_device(c.data_ptr(), c.numel())
        h.sync()
        row["chunk%d_ms" % i] = (time.perf_counter() - t0) * 1e3
        row["chunk%d_kernels_ms" % i] = [round(h.kernel_time(j)[0], 3) for j in range(4)]
    rows.append(row)
st = h.stats()
print(json.dumps({"geometry": "k=24, 2^26 slots, 2 x 2M x 100 bp", "instances": st["instances"], "distinct": st["distinct"],
                  "runs": rows[1:]}))
