#!/usr/bin/env python3
"""Early preparation on and off on ONE handle (so that both meet the same placement of its buffers, see DESIGN
section 6): a configs[2] step (four pieces of 12.5 M reads, 2^31 slots, one window pass). Round 3 also ran it with the
preparation of piece i+1 started at once / after level 1 of piece i / on a stream of lower or higher priority (the
switches are gone from the library; results in DESIGN section 6)."""
import os, sys, time
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
h.set_mode("partition")


def step():
    h.reset()
    for p in pieces:
        h.add_fastq_device(p.data_ptr(), p.numel())
    h.sync()


def timed(env, n=3):
    os.environ.pop("QH_NO_EARLY_PREP", None)
    os.environ.update(env)
    step()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    return (time.perf_counter() - t0) / n * 1e3


variants = [("early", {}), ("not early", {"QH_NO_EARLY_PREP": "1"})]
for rnd in range(3):
    print("  ".join("%s %.2f ms" % (name, timed(env)) for name, env in variants), flush=True)
