"""How much would overlapping the kernels of consecutive chunks buy? Two independent builds (own tables, own streams) fed the
same chunk in turn, against one build: the upper bound of what pipelining chunks inside one handle could reach."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quorum_amd as qa
from quorum_amd import synth

g = synth.make_genome(5_000_000)
c = torch.from_numpy(synth.make_fastq(g, 2_000_000, 100, seed=42)).cuda()
torch.cuda.synchronize()
hs = [qa.HashWithQuality(50_000_000, 48, 7, min_qual=38) for _ in range(2)]
out = {}
for nh in (1, 2):
    use = hs[:nh]
    for rep in range(3):  # warm-up
        for h in use:
            h.reset(); h.add_fastq_device(c.data_ptr(), c.numel())
    for h in use: h.sync()
    steps = 20
    t0 = time.perf_counter()
    for rep in range(steps):
        for h in use:
            h.reset(); h.add_fastq_device(c.data_ptr(), c.numel())
    for h in use: h.sync()
    out["ms_per_chunk_%d_handles" % nh] = (time.perf_counter() - t0) * 1e3 / (steps * nh)
print(json.dumps(out))
