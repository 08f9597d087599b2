#!/usr/bin/env python3
"""Times the receive side of the sharded build on one GPU: FASTQ -> items bucketed by owner shard
(qh_build_extract_partitioned_device) -> qh_build_add_hashed_device of ONE shard's bucket into that shard's table.
    python tools/hashed_insert_bench.py --shard-bits 3 --reads 6000000 [--total-size 2000000000]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from quorum_amd import synth_gpu, _lib
from quorum_amd.mer_database import HashWithQuality

ap = argparse.ArgumentParser()
ap.add_argument("--shard-bits", type=int, default=3)
ap.add_argument("--reads", type=int, default=6_000_000)
ap.add_argument("--total-size", type=int, default=2_000_000_000)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--mode", default="auto")
a = ap.parse_args()
K, BITS, Q, L = 24, 7, 38, 100
dev = torch.device("cuda", 0)
world = 1 << a.shard_bits
genome = synth_gpu.make_genome(100_000_000, dev, seed=42)
codes, qual, isN = synth_gpu.make_reads(genome, a.reads, L, seed=1)
fq = synth_gpu.fastq_text(codes, qual, isN, first_id=0)
del codes, qual, isN, genome
hx = HashWithQuality(1 << 13, 2 * K, BITS, min_qual=Q, device=0, shard_bits=a.shard_bits, shard_id=0)
h = HashWithQuality(a.total_size, 2 * K, BITS, min_qual=Q, device=0, shard_bits=a.shard_bits, shard_id=0)
h.set_mode(a.mode, -1)
cap = int(a.reads * (L - K + 1) / world * 1.25) + 4096
buckets = torch.empty((world, cap), dtype=torch.int64, device=dev)
counts = torch.zeros(world, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
_lib.check(_lib.lib.qh_build_extract_partitioned_device(hx._h, C.c_void_p(fq.data_ptr()), fq.numel(), C.c_void_p(buckets.data_ptr()),
                                                        cap, C.c_void_p(counts.data_ptr())))
torch.cuda.synchronize()
t_ex = time.perf_counter() - t0
n0 = int(counts[0].item())
out = {"shard_bits": a.shard_bits, "reads": a.reads, "items_of_shard0": n0, "extract_ms": t_ex * 1e3, "runs": []}
for rep in range(a.reps):
    h.reset()
    h.sync()
    h.enable_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h.add_hashed_device(buckets[0].data_ptr(), n0)
    h.sync()
    dt = time.perf_counter() - t0
    out["runs"].append({"wall_ms": dt * 1e3, "G_items_per_s": n0 / dt / 1e9,
                        "kernel_ms": [h.kernel_time(i)[0] for i in range(4)]})
out["slots"] = h.stats()["slots"]
print(json.dumps(out))
