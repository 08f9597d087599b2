#!/usr/bin/env python3
"""Level-1 fan-out of the configs[1] step on one handle: 13 window-id bits split as b1 + (13 - b1)."""
import time
import numpy as np
import torch
import quorum_amd as qa
from quorum_amd import synth

g = synth.make_genome(5_000_000, seed=42)
fq = torch.from_numpy(synth.make_fastq(g, 2_000_000, 100, err=0.01, n_rate=0.0005, seed=42)).to("cuda:0")
torch.cuda.synchronize()
h = qa.HashWithQuality(50_000_000, 48, 7, min_qual=38)
h.enable_timing(True)


def step():
    h.reset()
    h.add_fastq_device(fq.data_ptr(), fq.numel())
    h.sync()


def timed(b1, n=20):
    h.set_mode("partition", b1)
    for _ in range(3):
        step()
    k0 = [h.kernel_time(i)[0] for i in range(4)]
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    dt = (time.perf_counter() - t0) / n * 1e3
    k1 = [h.kernel_time(i)[0] for i in range(4)]
    return dt, [(b - a) / n for a, b in zip(k0, k1)]


for rnd in range(2):
    for b1 in (7, 6, 5, 8, 4):
        dt, ks = timed(b1)
        print("b1 = %2d: step %.3f ms; segments %.3f, level 1 %.3f, level 2 %.3f, windows %.3f" % (b1, dt, *ks), flush=True)
