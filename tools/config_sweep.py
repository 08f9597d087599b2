"""Single-GPU build rate at a few other geometries (same timed region as bench.py: chunk resident in
HBM -> all k-mers in the table), to see the kernels that the headline configuration does not use."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quorum_amd as qa
from quorum_amd import synth

rows = []
for k, L, reads, size, bits in [(24, 100, 2_000_000, 50_000_000, 7), (31, 150, 1_500_000, 200_000_000, 7),
                                (31, 150, 1_500_000, 60_000_000, 7), (17, 250, 800_000, 50_000_000, 4),
                                (24, 100, 8_000_000, 200_000_000, 7)]:
    g = synth.make_genome(5_000_000 if reads <= 2_000_000 else 20_000_000)
    fq = synth.make_fastq(g, reads, L)
    chunk = torch.from_numpy(fq).to("cuda:0")
    h = qa.HashWithQuality(size, 2 * k, bits, min_qual=38)
    h.enable_timing(True)
    ts = []
    for it in range(4):
        h.reset(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        h.add_fastq_device(chunk.data_ptr(), chunk.numel()); h.sync()
        ts.append(time.perf_counter() - t0)
    st = h.stats()
    kt = [h.kernel_time(i) for i in range(4)]
    rows.append({"k": k, "L": L, "reads": reads, "size": size, "bits": bits, "instances": st["instances"],
                 "distinct": st["distinct"], "slots": st["slots"], "grows": st["grows"], "ms": min(ts[1:]) * 1e3,
                 "G_kmers_per_s": st["instances"] / min(ts[1:]) / 1e9,
                 "kernel_ms": [round(ms / n, 3) if n else 0 for ms, n in kt]})
    print(json.dumps(rows[-1]), flush=True)
    h.close(); del chunk
