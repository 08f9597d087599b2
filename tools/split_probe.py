#!/usr/bin/env python3
"""Level-1 fan-out of the configs[2] step on ONE handle (same buffers for every setting):
18 window-id bits split as b1 + (18 - b1)."""
import sys, time
import torch
import quorum_amd as qa
from quorum_amd import synth_gpu

dev = torch.device("cuda:0")
genome = synth_gpu.make_genome(100_000_000, dev, seed=42)
pieces = []
for i in range(4):
    codes, qual, isN = synth_gpu.make_reads(genome, 12_500_000, 100, seed=i + 1)
    pieces.append(synth_gpu.fastq_text(codes, qual, isN, first_id=i * 12_500_000))
    del codes, qual, isN
del genome
torch.cuda.empty_cache()
torch.cuda.synchronize()
h = qa.HashWithQuality(2_000_000_000, 48, 7, min_qual=38)
h.enable_timing(True)


def step():
    h.reset()
    for p in pieces:
        h.add_fastq_device(p.data_ptr(), p.numel())
    h.sync()


def timed(b1, n=3):
    h.set_mode("partition", b1)
    step()
    k0 = [h.kernel_time(i)[0] for i in range(4)]
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    dt = (time.perf_counter() - t0) / n * 1e3
    k1 = [h.kernel_time(i)[0] for i in range(4)]
    return dt, [(b - a) / n for a, b in zip(k0, k1)]


for rnd in range(2):
    for b1 in (9, 8, 10, 7, 9):
        dt, ks = timed(b1)
        print("b1 = %2d: step %.2f ms; per step: segments %.2f, level 1 %.2f, level 2 %.2f, windows %.2f" % (b1, dt, *ks), flush=True)
