"""Single-GPU build rate at a few other geometries (same timed region as bench.py: chunk resident in
HBM -> all k-mers in the table), to see the kernels that the headline configuration does not use."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quorum_amd as qa
from quorum_amd import synth

rows = []
for k, L, reads, size, bits in [] if "--only-big" in sys.argv else [(24, 100, 2_000_000, 50_000_000, 7), (31, 150, 1_500_000, 200_000_000, 7),
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

# BASELINE configs[3] geometry on one GPU: k = 31, 150 bp, a 2^32-slot table (34 GB; the counting window kernel with wide
# tags, 13-bit counters), one chunk of 12 M reads drawn on the GPU (3.8 GB of text, 1.4 G k-mers)
if "--big" in sys.argv:
    from quorum_amd import synth_gpu
    dev = torch.device("cuda:0")
    k, L, reads, size, bits = 31, 150, 12_000_000, 1 << 32, 7
    genome = synth_gpu.make_genome(400_000_000, dev, seed=7)
    parts = []
    for p0 in range(0, reads, 2_000_000):
        codes, qual, isN = synth_gpu.make_reads(genome, 2_000_000, L, seed=100 + p0)
        parts.append(synth_gpu.fastq_text(codes, qual, isN, first_id=p0))
        del codes, qual, isN
    chunk = torch.cat(parts); del parts, genome
    torch.cuda.synchronize()
    h = qa.HashWithQuality(size, 2 * k, bits, min_qual=38)
    h.enable_timing(True)
    ts = []
    for it in range(3):
        h.reset(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        h.add_fastq_device(chunk.data_ptr(), chunk.numel()); h.sync()
        ts.append(time.perf_counter() - t0)
    st = h.stats()
    kt = [h.kernel_time(i) for i in range(4)]
    print(json.dumps({"k": k, "L": L, "reads": reads, "size": size, "bits": bits, "instances": st["instances"],
                      "distinct": st["distinct"], "slots": st["slots"], "grows": st["grows"], "ms": min(ts[1:]) * 1e3,
                      "G_kmers_per_s": st["instances"] / min(ts[1:]) / 1e9,
                      "kernel_ms": [round(ms / n, 3) if n else 0 for ms, n in kt]}), flush=True)
    h.close()
