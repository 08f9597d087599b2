#!/usr/bin/env python3
"""Why does the level-1 split of a configs[2] piece take 4.8 ms in the first process on a fresh box and 5.6 ms in later
ones? One piece of 12.5 M reads into a 2^31-slot table, level-1 time per handle: a first handle, a second one after
the first was closed, a third after the process churned 180 GB of device memory in odd sizes. Run it twice in a row."""
import sys, time
import torch
import quorum_amd as qa
from quorum_amd import synth_gpu

dev = torch.device("cuda:0")
genome = synth_gpu.make_genome(100_000_000, dev, seed=42)
codes, qual, isN = synth_gpu.make_reads(genome, 12_500_000, 100, seed=1)
piece = synth_gpu.fastq_text(codes, qual, isN, first_id=0)
del codes, qual, isN, genome
torch.cuda.empty_cache()
torch.cuda.synchronize()


def probe(tag):
    h = qa.HashWithQuality(2_000_000_000, 48, 7, min_qual=38)
    h.set_mode("partition")
    h.set_defer(0)
    h.enable_timing(True)
    last = [h.kernel_time(i) for i in range(4)]
    out = []
    for rep in range(4):
        h.reset()
        h.add_fastq_device(piece.data_ptr(), piece.numel())
        h.sync()
        now = [h.kernel_time(i) for i in range(4)]
        out.append(tuple(round(now[i][0] - last[i][0], 3) for i in range(4)))
        last = now
    print(tag, "prep / level 1 / level 2 / windows (ms):", out[1:], flush=True)
    h.close()


probe("first handle     ")
probe("second handle    ")
junk = [torch.empty(int(2.7e9) + 4096 * 37 * i, dtype=torch.uint8, device=dev) for i in range(60)]
del junk[::2]
torch.cuda.empty_cache()
probe("fragmented       ")
del junk
torch.cuda.empty_cache()
probe("after freeing all")
