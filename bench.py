#!/usr/bin/env python3
"""bench.py -- k-mer insertions/s of the MI355X database build (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step = one complete database build over one batch of synthetic reads whose FASTQ
text is already resident in HBM: table clear + FASTQ indexing + k-mer extraction +
(N>1: all-to-all exchange) + insertion, until every k-mer is visible in the table.
N=1 workload = BASELINE.json configs[1]: 5 Mbp genome, 2M x 100 bp reads, 1 % errors,
k=24, -s 50M, -b 7, -q 38 (154 M insertions per step).
N>1 workload = BASELINE.json configs[2]: 100 Mbp genome, 50 M x 100 bp reads IN ALL (3.85 G insertions
per step), k=24, -s 2G hash-sharded over the N GPUs: strong scaling, every rank holds 50M / N reads in
pieces of 2 M that flow through the extract / exchange / insert pipeline (quorum_amd/dist_build.py).
`--workload c2` runs that workload on one GPU too (the matching N=1 point of the scaling curve).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

K, BITS, MIN_QUAL, READ_LEN = 24, 7, 38, 100
GENOME_PER_GPU, READS_PER_GPU, SIZE_PER_GPU = 5_000_000, 2_000_000, 50_000_000                 # configs[1]
C2_GENOME, C2_READS_TOTAL, C2_SIZE, C2_PIECE_READS = 100_000_000, 50_000_000, 2_000_000_000, 2_000_000  # configs[2]
ALG_BYTES_PER_KMER = 16.0 + 2.0 * READ_LEN / (READ_LEN - K + 1)   # 18.6 B
HBM_PEAK_GBS = 8000.0                                               # MI355X_MICROARCH.md: 8 TB/s


def cpu_baseline(fq, n_reads_total):
    """The oracle's multithreaded CPU restatement of the reference algorithm, timed on a bounded
    sample (up to the first 2M reads = one bench chunk, ~25 core-seconds). Checker code: never the thing
    measured as `value`."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    from quorum_amd import synth
    sample_reads = min(n_reads_total, 2_000_000)
    nbytes = sample_reads * synth.record_bytes(READ_LEN)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1)
    threads = max(1, min(threads, os.cpu_count() or 1))
    db = po.OracleDB(SIZE_PER_GPU, K, BITS)
    buf = np.ascontiguousarray(fq[:nbytes])
    t0 = time.perf_counter()
    n = db.add_fastq_ptr(buf.ctypes.data, nbytes, MIN_QUAL, threads)
    dt = time.perf_counter() - t0
    db.close()
    return {"value": n / dt, "unit": "k-mers/s", "cores": threads, "kind": "port",
            "sample": "first %d of the %d reads (%d insertions, %.2f s): oracle/quorum_oracle.c, "
                      "multithreaded CPU restatement of the reference algorithm (not the upstream binary)"
                      % (sample_reads, n_reads_total, n, dt),
            "cpu_seconds": dt * threads}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: the named config)")
    ap.add_argument("--workload", default="auto", choices=["auto", "c1", "c2"],
                    help="auto: BASELINE configs[1] on one GPU, configs[2] (50 M reads in all) on several")
    ap.add_argument("--piece-reads", type=int, default=0,
                    help="configs[2]: reads per pipelined piece (default: so that a piece brings a shard about half as many "
                         "k-mers as it has slots -- the window pass streams the whole shard once per piece -- but at least 2 M)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="auto", choices=["auto", "direct", "partition"], help="build strategy (A/B runs)")
    ap.add_argument("--level1-bits", type=int, default=-1)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    # QH_DIST_BACKEND=gloo + QH_SHARE_GPU=1: rehearsal of the N>1 path with all ranks on GPU 0
    backend = os.environ.get("QH_DIST_BACKEND", "nccl")
    if os.environ.get("QH_SHARE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # QH_FORCE_PIPELINE=1 (under torch.distributed.run --nproc-per-node 1): the one rank takes the N>1 route, process group,
    # RCCL all-to-all (to itself) and all -- the rehearsal of that route a one-GPU box allows
    pipe = bool(os.environ.get("QH_FORCE_PIPELINE")) and world == 1
    grouped = world > 1 or pipe
    if grouped:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from quorum_amd import synth
    from quorum_amd.dist_build import ShardedBuilder

    workload = args.workload if args.workload != "auto" else ("c1" if world == 1 else "c2")
    if workload == "c1":
        reads_rank = args.reads or READS_PER_GPU
        genome_bases, total_size = GENOME_PER_GPU * world, SIZE_PER_GPU * world
        genome = synth.make_genome(genome_bases, seed=42)
        fq = synth.make_fastq(genome, reads_rank, READ_LEN, err=0.01, n_rate=0.0005, seed=42, first_id=rank * reads_rank)
        pieces = [torch.from_numpy(fq).to(dev)]          # the FASTQ text, resident in HBM
        piece_reads = [reads_rank]
    else:
        # configs[2]: the reads are drawn on the GPU (50 M reads are too many to draw on the host in a benchmark run)
        from quorum_amd import synth_gpu
        reads_rank = args.reads or C2_READS_TOTAL // world
        genome_bases, total_size = C2_GENOME, C2_SIZE
        genome = synth_gpu.make_genome(genome_bases, dev, seed=42)
        pieces, piece_reads, fq = [], [], None
        if not args.piece_reads:
            slots_rank = (1 << (C2_SIZE - 1).bit_length()) // world
            want = max(C2_PIECE_READS, slots_rank // 2 // (READ_LEN - K + 1))
            args.piece_reads = -(-reads_rank // max(1, round(reads_rank / want)))   # equal pieces
        for p0 in range(0, reads_rank, args.piece_reads):
            n = min(args.piece_reads, reads_rank - p0)
            codes, qual, isN = synth_gpu.make_reads(genome, n, READ_LEN, seed=1000 * rank + len(pieces) + 1)
            pieces.append(synth_gpu.fastq_text(codes, qual, isN, first_id=rank * reads_rank + p0))
            piece_reads.append(n)
            del codes, qual, isN
        del genome
        torch.cuda.empty_cache()
    torch.cuda.synchronize()

    b = ShardedBuilder(K, BITS, total_size, MIN_QUAL, rank, world, local, force_pipeline=pipe)
    b.h.set_mode(args.mode, args.level1_bits)
    b.h.enable_timing(True)

    def step():
        b.reset()
        for piece, n in zip(pieces, piece_reads):
            b.add_fastq_device(piece, max_items=n * (READ_LEN - K + 1))
        b.flush()
        b.h.sync()

    def fence():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    st0 = b.h.stats()
    b.h.enable_timing(True)                         # restart the kernel-time accumulation
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    ktimes = [b.h.kernel_time(i) for i in range(4)]
    st = b.h.stats()

    rdev = dev if backend == "nccl" else torch.device("cpu")
    t = torch.tensor([dt], dtype=torch.float64, device=rdev)
    inst = torch.tensor([float(st["instances"])], dtype=torch.float64, device=rdev)
    if grouped:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(inst, op=dist.ReduceOp.SUM)
    dt = float(t.item())
    per_step = float(inst.item())                   # insertions of ONE step over all ranks (stats are reset each step)

    if rank == 0:
        # HIP-event averages of the main kernels (recorded on the stream they are launched on)
        avgs = [ms / n if n else 0.0 for ms, n in ktimes]
        partitioned = ktimes[3][1] > 0
        # names as rocprofv3 prints them: k_partition<0, ...> = FASTQ source (level 1), k_partition<1, ...> = item source
        # (owner check; class of k: 0 is k >= 12)
        single = not grouped
        knames = ("segments(k_rec_segs_sum+k_scan_u32+k_rec_segs_apply)" if partitioned and single
                  else "k_build_direct" if single else "k_insert_items",
                  "k_partition<0, false, 0>" if single else "k_partition<1, false, 0> (level 1)",
                  "k_partition<1, false, 0>" if single else "k_partition<1, false, 0> (level 2)",
                  "k_window_count<true>")  # the counting form with 16-bit counters at this geometry
        step_s = dt / args.steps
        per_gpu_step = per_step / world             # insertions per GPU and step (the roofline is a per-GPU quantity)
        launches_per_step = max(1.0, ktimes[3][1] / args.steps) if partitioned else 1.0   # one per piece of reads
        per_launch = st["instances"] / launches_per_step   # this rank's insertions per launch of each main kernel
        # Headline roofline = the WHOLE STEP: SURVEY 8(d)'s 18.6 algorithmic bytes per k-mer (16 B of table traffic + the
        # read's text) over the step's wall time. No single kernel moves those bytes (level 1 never touches the table,
        # the window pass never sees the text), so pricing one kernel's time with them would flatter it.
        achieved = per_gpu_step * ALG_BYTES_PER_KMER / step_s / 1e9
        # Each main kernel against the bytes IT must move per k-mer in this three-pass design:
        text_b = 2.0 * READ_LEN / (READ_LEN - K + 1)
        slots_per_kmer = st["slots"] / max(1.0, per_launch)
        own_bytes = (None,                                  # slot 0: small preparation kernels (or the direct path)
                     text_b + 8.0,                          # level 1: the read's text in, one 8-byte item out
                     16.0,                                  # level 2: item in, item out
                     8.0 + 8.0 * slots_per_kmer)            # windows: item in, every slot of the table written once
        dom = max(range(1 if partitioned else 0, 4), key=lambda i: avgs[i])
        per_kernel = {}
        for i in range(4):
            if avgs[i] > 0:
                e = {"avg_ms": avgs[i], "launches": int(ktimes[i][1])}
                if partitioned and own_bytes[i]:
                    e["own_alg_bytes_per_kmer"] = own_bytes[i]
                    e["own_GBps"] = per_launch * own_bytes[i] / (avgs[i] * 1e-3) / 1e9
                    e["own_frac_of_hbm_peak"] = e["own_GBps"] / HBM_PEAK_GBS
                per_kernel[knames[i]] = e
        # PMC traffic (FETCH_SIZE x2 + WRITE_SIZE, separate passes: tools/profile_bench.sh) of all kernels of one step
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if world == 1 and workload == "c1" and not pipe and not args.reads and os.path.exists(pmc):   # measured on this workload only
            try:
                traffic = json.load(open(pmc)).get("whole_step_hbm_bytes")
            except Exception:
                traffic = None
        out = {
            "metric": "k-mers/s into HQ-bit hash", "value": per_step * args.steps / dt, "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_s * 1e3,
            "higher_is_better": True, "scaling": "weak" if workload == "c1" else "strong", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: synthetic %d Mbp genome, %d x %d bp reads per GPU, 1%% subst errors, "
                                    "k=%d, -s %dM, -b %d, -q %d; FASTQ resident in HBM; step = clear + index + extract + %sinsert"
                                    % (genome_bases // 1_000_000, reads_rank, READ_LEN, K, total_size // 1_000_000, BITS, MIN_QUAL,
                                       "all-to-all exchange + " if world > 1 else "")) if workload == "c1" else
                                   ("BASELINE configs[2]: synthetic %d Mbp genome, %d x %d bp reads IN ALL (%d per GPU in pieces of "
                                    "%d), 1%% subst errors, k=%d, -s %dM sharded by hash over %d GPU(s), -b %d, -q %d; FASTQ resident "
                                    "in HBM; step = clear + per piece (index + extract by owner%s + insert), pieces pipelined"
                                    % (genome_bases // 1_000_000, reads_rank * world, READ_LEN, reads_rank, args.piece_reads, K,
                                       total_size // 1_000_000, world, BITS, MIN_QUAL,
                                       " + all-to-all exchange" if grouped else "")),
                       "insertions_per_step": int(per_step), "distinct_per_gpu": int(st["distinct"]),
                       "slots_per_gpu": int(st["slots"]), "pieces_per_gpu": len(pieces),
                       "bucket_overflow_retries": int(getattr(b, "retries", 0)),
                       "exchange_rounds_max": int(getattr(b, "rounds_max", 0)), "parallelism": "hash-sharded x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "scope": "whole step (every kernel of one build): algorithmic bytes / step time",
                         "alg_bytes_per_kmer": ALG_BYTES_PER_KMER, "kmers_per_launch": int(per_launch),
                         "dominant_kernel": knames[dom], "kernel_avg_ms": avgs[dom], "kernel_launches": int(ktimes[dom][1]),
                         "kernels": per_kernel},
        }
        if world == 1 and workload == "c1":
            # measured random-access roofline: uniformly random 64-bit atomicCAS / load+CAS /
            # 8-byte loads over a table of this config's size, same number of updates as one step
            import ctypes as C
            from quorum_amd._lib import lib, check
            lsize = int(st["slots"]).bit_length() - 1
            ra = {}
            for name, mode in (("cas", 0), ("load_then_cas", 2), ("load8", 1)):
                ms = C.c_double()
                check(lib.qh_bench_random_access(local, lsize, int(per_launch), mode, 3, C.byref(ms)))
                ra[name + "_per_s"] = per_launch / (ms.value * 1e-3)
            # not a roofline fraction: how the partitioned build's whole-step rate compares with what one global
            # atomic per k-mer could reach at best
            ra["step_kmers_per_s"] = out["value"]
            ra["step_vs_random_cas"] = out["value"] / ra["cas_per_s"]
            out["random_access_roofline"] = ra
        if world == 1 and not args.no_cpu_baseline:
            if fq is None:
                fq = pieces[0].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(fq, piece_reads[0])
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
