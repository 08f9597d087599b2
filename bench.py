#!/usr/bin/env python3
"""bench.py -- k-mer insertions/s of the MI355X database build (BASELINE.json metric: "k-mers/s into HQ-bit hash at
1/2/4/8 GPUs").

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step = one complete database build over one batch of synthetic reads whose FASTQ text is already resident in HBM:
table clear + FASTQ indexing + k-mer extraction + (N>1: all-to-all exchange) + insertion, until every k-mer is visible
in the table (the step ends with qh_build_sync, which applies whatever the deferred window pass still holds).

EVERY N runs the same workload, so that the driver's 1/2/4/8 run is one strong-scaling curve: BASELINE.json configs[2]
-- 100 Mbp genome, 50 M x 100 bp reads IN ALL (3.85 G insertions per step), 1 % errors, k=24, -s 2G hash-sharded over the
N GPUs, -b 7, -q 38; every rank holds 50M / N reads in pieces that flow through the extract / exchange / insert pipeline
(quorum_amd/dist_build.py). At N = 1 the line also carries, under "configs1", the figures of configs[1] (5 Mbp genome,
2 M reads, -s 50M: the headline of rounds 1 and 2) measured in the same run, so that rounds stay comparable;
`--workload c1` makes that workload the line's own.

Before the timed region every run checks itself: a small prefix of every rank's reads goes through the very path that
is timed and the sharded table's content (entries, and an XOR over every (mer, value) pair) must equal that of a
single-table build of the same reads made by the direct strategy; at N > 1 a 1.6 GB message is sent between ranks 2i
and 2i+1 and checked word by word ("large_message_ok"; the exchange cuts its messages at "exchange_message_cap" bytes
whatever the answer). Rank 0 prints ONE JSON line.
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
C1_GENOME, C1_READS, C1_SIZE = 5_000_000, 2_000_000, 50_000_000                                     # configs[1]
C2_GENOME, C2_READS_TOTAL, C2_SIZE, C2_PIECE_READS = 100_000_000, 50_000_000, 2_000_000_000, 2_000_000  # configs[2]
ALG_BYTES_PER_KMER = 16.0 + 2.0 * READ_LEN / (READ_LEN - K + 1)   # 18.6 B
HBM_PEAK_GBS = 8000.0                                               # MI355X_MICROARCH.md: 8 TB/s
KPR = READ_LEN - K + 1                                              # k-mers per read


def cpu_baseline(fq, n_reads_total, table_size):
    """The oracle's multithreaded CPU restatement of the reference algorithm, timed on a bounded
    sample (up to the first 2M reads, ~25 core-seconds). Checker code: never the thing measured as `value`."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    from quorum_amd import synth
    sample_reads = min(n_reads_total, 2_000_000)
    nbytes = sample_reads * synth.record_bytes(READ_LEN)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1)
    threads = max(1, min(threads, os.cpu_count() or 1))
    db = po.OracleDB(table_size, K, BITS)
    buf = np.ascontiguousarray(fq[:nbytes])
    t0 = time.perf_counter()
    n = db.add_fastq_ptr(buf.ctypes.data, nbytes, MIN_QUAL, threads)
    dt = time.perf_counter() - t0
    db.close()
    return {"value": n / dt, "unit": "k-mers/s", "cores": threads, "kind": "port",
            "sample": "first %d of this rank's %d reads into a table of %d slots (%d insertions, %.2f s): "
                      "oracle/quorum_oracle.c, multithreaded CPU restatement of the reference algorithm (not the upstream "
                      "binary)" % (sample_reads, n_reads_total, table_size, n, dt),
            "cpu_seconds": dt * threads}


def content_digest(mers, vals):
    """(entries, XOR over a mix of every (mer, value) pair) of a table dump: order free, so shards combine by XOR"""
    if mers.size == 0:
        return 0, 0
    x = (mers * np.uint64(0x9E3779B97F4A7C15)) ^ (vals * np.uint64(0xC2B2AE3D27D4EB4F)) ^ (mers >> np.uint64(29))
    return int(mers.size), int(np.bitwise_xor.reduce(x))


def make_c1(rank, world, dev, reads):
    from quorum_amd import synth
    genome = synth.make_genome(C1_GENOME * world, seed=42)
    fq = synth.make_fastq(genome, reads, READ_LEN, err=0.01, n_rate=0.0005, seed=42, first_id=rank * reads)
    return [torch.from_numpy(fq).to(dev)], [reads], fq


def make_c2(rank, world, dev, reads_rank, piece_reads):
    """the reads are drawn on the GPU (50 M reads are too many to draw on the host in a benchmark run)"""
    from quorum_amd import synth_gpu
    genome = synth_gpu.make_genome(C2_GENOME, dev, seed=42)
    pieces, sizes = [], []
    for p0 in range(0, reads_rank, piece_reads):
        n = min(piece_reads, reads_rank - p0)
        codes, qual, isN = synth_gpu.make_reads(genome, n, READ_LEN, seed=1000 * rank + len(pieces) + 1)
        pieces.append(synth_gpu.fastq_text(codes, qual, isN, first_id=rank * reads_rank + p0))
        sizes.append(n)
        del codes, qual, isN
    del genome
    torch.cuda.empty_cache()
    return pieces, sizes


def measure(b, pieces, piece_reads, steps, warmup, fence):
    """-> (seconds of `steps` steps, kernel times, stats of the last step)"""
    def step():
        # (every step is collected before the next begins. Enqueued back to back instead -- reset() notes itself while
        #  work is in flight, flush() starts the pass over the table without waiting -- the steps take the same time:
        #  the pass over the table keeps every CU's LDS, nothing of the next step can run beside it)
        b.reset()
        for piece, n in zip(pieces, piece_reads):
            b.add_fastq_device(piece, max_items=n * KPR)
        b.flush()
        b.h.sync()

    for _ in range(warmup):
        step()
    b.h.enable_timing(True)                         # (re)start the kernel-time accumulation
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    return dt, [b.h.kernel_time(i) for i in range(4)], b.h.stats()


def kernel_table(ktimes, st, steps, single, per_rank_step, shard_bits=0):
    """HIP-event averages of the main kernels (recorded on the stream they are launched on), each against the bytes
    IT must move per k-mer in this three-pass design"""
    avgs = [ms / n if n else 0.0 for ms, n in ktimes]
    partitioned = ktimes[3][1] > 0
    # window buckets hold 4-byte items when remainder, position and quality bit fit (qh_items_pack_gsize)
    item_b = 4.0 if 2 * K + 1 - (int(st["slots"]).bit_length() - 1) - shard_bits + 13 + 1 <= 32 else 8.0
    # the split of the window-id bits as run_partitioned makes it (qh_build.hip): level 2 takes 18-block tiles at 9 or 10 bits
    wid = int(st["slots"]).bit_length() - 1 - 13
    b1 = wid if wid <= 8 else min((wid + 1) // 2, max(8, wid - 10))
    b2 = wid - b1
    knames = ("segments(k_rec_segs_sum+k_scan_u32_sums/top/apply+k_rec_segs_apply)" if partitioned and single
              else "k_build_direct" if single else "k_insert_items",
              # (the instantiations rocprofv3 lists for these geometries: k = 24, bin field in the low 32 bits, tile
              #  blocks and bucket limit; the shard handle of the pipeline checks the owner of what it receives)
              "k_partition<0, false, 0, 24, true, 12, 4096>" if single else "k_partition<1, true, 0, 0, false, 12, 4096> (level 1)",
              "k_partition<1, false, 0, 0, true, %s>%s" % ("18, 1024" if 9 <= b2 <= 10 else "12, 4096", "" if single else " (level 2)"),
              # the counting form with 16-bit counters at these geometries, over packed or 8-byte items
              "k_window_count<true, false, true>" if item_b == 4.0 else "k_window_count<true, false, false>")
    text_b = 2.0 * READ_LEN / KPR
    per_kernel = {}
    for i in range(4):
        if avgs[i] <= 0:
            continue
        launches_per_step = ktimes[i][1] / steps
        per_launch = per_rank_step / max(launches_per_step, 1e-9)   # this rank's insertions per launch of the kernel
        e = {"avg_ms": avgs[i], "launches": int(ktimes[i][1]), "launches_per_step": launches_per_step}
        own = (None, text_b + 8.0, 8.0 + item_b, item_b + 8.0 * st["slots"] / max(1.0, per_launch))[i] if partitioned else None
        if own:
            # level 1: the read's text in, one 8-byte item out; level 2: item in, item out; windows: item in, every
            # slot of the table written once per pass
            e["own_alg_bytes_per_kmer"] = own
            e["own_GBps"] = per_launch * own / (avgs[i] * 1e-3) / 1e9
            e["own_frac_of_hbm_peak"] = e["own_GBps"] / HBM_PEAK_GBS
        per_kernel[knames[i]] = e
    # the kernel that takes most of a step (time per launch x launches per step)
    dom = max(range(1 if partitioned else 0, 4), key=lambda i: avgs[i] * ktimes[i][1])
    return per_kernel, knames[dom], avgs[dom], int(ktimes[dom][1])


def roofline_block(value_per_gpu, ktimes, st, steps, single, per_rank_step, traffic, shard_bits=0):
    achieved = value_per_gpu * ALG_BYTES_PER_KMER / 1e9
    per_kernel, dom, dom_ms, dom_n = kernel_table(ktimes, st, steps, single, per_rank_step, shard_bits)
    # Headline roofline = the WHOLE STEP: SURVEY 8(d)'s 18.6 algorithmic bytes per k-mer (16 B of table traffic + the
    # read's text) over the step's wall time. No single kernel moves those bytes (level 1 never touches the table,
    # the window pass never sees the text), so pricing one kernel's time with them would flatter it.
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "scope": "whole step (every kernel of one build): algorithmic bytes / step time",
            "alg_bytes_per_kmer": ALG_BYTES_PER_KMER, "kmers_per_step_per_gpu": int(per_rank_step),
            "dominant_kernel": dom, "kernel_avg_ms": dom_ms, "kernel_launches": dom_n, "kernels": per_kernel}


def pmc_traffic(key):
    """PMC traffic (FETCH_SIZE x2 + WRITE_SIZE, separate passes: tools/profile_bench.sh) of all kernels of one step"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(key)
    except Exception:
        return None


def large_message_check(rank, world, dev):
    """1.6 GB from rank r to rank r ^ 1 (to itself when alone) in one list-form all-to-all, checked word by word"""
    n = int(os.environ.get("QH_BENCH_BIG_MESSAGE_ITEMS", 200_000_000))
    partner = rank ^ 1 if world > 1 and (rank ^ 1) < world else rank

    def payload(src, dst):
        return torch.arange(n, dtype=torch.int64, device=dev) * 2654435761 + (src * 1000003 + dst * 7919 + 12345)

    empty = torch.empty(0, dtype=torch.int64, device=dev)
    send = [payload(rank, partner) if d == partner else empty for d in range(world)]
    got = torch.zeros(n, dtype=torch.int64, device=dev)
    recv = [got if s == partner else torch.empty(0, dtype=torch.int64, device=dev) for s in range(world)]
    dist.all_to_all(recv, send)
    torch.cuda.synchronize()
    bad = (got != payload(partner, rank)).sum().to(torch.int64)
    dist.all_reduce(bad)
    return {"bytes": n * 8, "mismatched_words_all_ranks": int(bad.item())}


def content_check(b, pieces, piece_reads, rank, world, dev, grouped, backend, total_size):
    """A prefix of every rank's reads through the path that is timed -> entries and XOR digest of the (sharded) table,
    against a single-table build of the same reads by the direct strategy (rank 0)"""
    import quorum_amd as qa
    from quorum_amd import synth
    n = min(100_000, min(piece_reads))
    nbytes = n * synth.record_bytes(READ_LEN)
    prefix = pieces[0][:nbytes]
    b.reset()
    b.add_fastq_device(prefix, max_items=n * KPR)
    st = b.finish()
    entries, digest = content_digest(*b.h.dump())
    inst = int(st["instances"])
    if grouped:
        cpu = backend != "nccl"
        signed = int(np.array([digest], dtype=np.uint64).view(np.int64)[0])
        t = torch.tensor([entries, inst, signed], dtype=torch.int64, device="cpu" if cpu else dev)
        rows = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(rows, t)
        entries = sum(int(r[0]) for r in rows)
        inst = sum(int(r[1]) for r in rows)
        digest = 0
        for r in rows:
            digest ^= int(np.array([int(r[2])], dtype=np.int64).view(np.uint64)[0])
        src = prefix.cpu() if cpu else prefix
        texts = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(texts, src)
    else:
        texts = [prefix]
    out = None
    if rank == 0:
        ref = qa.HashWithQuality(max(1 << 20, 2 * n * world * KPR), 2 * K, BITS, min_qual=MIN_QUAL, device=dev.index)
        ref.set_mode("direct")
        ref_inst = 0
        for t in texts:
            t = t.to(dev)
            torch.cuda.synchronize()
            ref.add_fastq_device(t.data_ptr(), t.numel())
            ref.sync()
        ref_inst = int(ref.done()["instances"])
        r_entries, r_digest = content_digest(*ref.dump())
        ref.close()
        out = {"reads_per_rank": n, "insertions": inst, "entries": entries, "xor": "%016x" % digest,
               "reference": "single table, direct global-CAS strategy, same reads (rank 0)",
               "ok": bool(entries == r_entries and digest == r_digest and inst == ref_inst)}
    del texts
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: the named config)")
    ap.add_argument("--workload", default="c2", choices=["auto", "c1", "c2"],
                    help="c2 (default, same as auto): BASELINE configs[2] at every N; c1: configs[1] per GPU")
    ap.add_argument("--piece-reads", type=int, default=0,
                    help="configs[2]: reads per piece (default: 12.5 M on one GPU, where a piece is one chunk of the deferred "
                         "window pass; 2 M in the exchange pipeline of several)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c1", action="store_true", help="N = 1: leave the configs[1] figures out")
    ap.add_argument("--no-checks", action="store_true", help="skip the content and large-message checks (profiling runs)")
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

    from quorum_amd.dist_build import ShardedBuilder, _max_message_items

    workload = "c2" if args.workload == "auto" else args.workload
    fq = None
    if workload == "c1":
        reads_rank = args.reads or C1_READS
        genome_bases, total_size = C1_GENOME * world, C1_SIZE * world
        pieces, piece_reads, fq = make_c1(rank, world, dev, reads_rank)
    else:
        reads_rank = args.reads or C2_READS_TOTAL // world
        genome_bases, total_size = C2_GENOME, C2_SIZE
        if not args.piece_reads:
            if grouped:
                slots_rank = (1 << (C2_SIZE - 1).bit_length()) // world
                want = max(C2_PIECE_READS, slots_rank // 8 // KPR)
            else:
                want = 12_500_000
            args.piece_reads = -(-reads_rank // max(1, round(reads_rank / want)))   # equal pieces
        pieces, piece_reads = make_c2(rank, world, dev, reads_rank, args.piece_reads)
    torch.cuda.synchronize()

    def fence():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    big = None
    if grouped and backend == "nccl" and not args.no_checks:
        big = large_message_check(rank, world, dev)

    # (the reads are resident and the device synchronised before anything is timed: the handle keeps its own streams)
    b = ShardedBuilder(K, BITS, total_size, MIN_QUAL, rank, world, local, force_pipeline=pipe, own_stream=True)
    b.h.set_mode(args.mode, args.level1_bits)
    check = None if args.no_checks else content_check(b, pieces, piece_reads, rank, world, dev, grouped, backend, total_size)
    fence()

    dt, ktimes, st = measure(b, pieces, piece_reads, args.steps, args.warmup, fence)

    rdev = dev if backend == "nccl" else torch.device("cpu")
    t = torch.tensor([dt], dtype=torch.float64, device=rdev)
    inst = torch.tensor([float(st["instances"])], dtype=torch.float64, device=rdev)
    if grouped:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(inst, op=dist.ReduceOp.SUM)
    dt = float(t.item())
    per_step = float(inst.item())                   # insertions of ONE step over all ranks (stats are reset each step)

    if rank == 0:
        single = not grouped
        step_s = dt / args.steps
        value = per_step * args.steps / dt
        full = not args.reads and not pipe
        traffic = pmc_traffic("whole_step_hbm_bytes" if workload == "c1" else "whole_step_hbm_bytes_c2") if world == 1 and full else None
        out = {
            "metric": "k-mers/s into HQ-bit hash", "value": value, "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_s * 1e3,
            "higher_is_better": True, "scaling": "weak" if workload == "c1" else "strong", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: synthetic %d Mbp genome, %d x %d bp reads per GPU, 1%% subst errors, "
                                    "k=%d, -s %dM, -b %d, -q %d; FASTQ resident in HBM; step = clear + index + extract + %sinsert"
                                    % (genome_bases // 1_000_000, reads_rank, READ_LEN, K, total_size // 1_000_000, BITS, MIN_QUAL,
                                       "all-to-all exchange + " if world > 1 else "")) if workload == "c1" else
                                   ("BASELINE configs[2]: synthetic %d Mbp genome, %d x %d bp reads IN ALL (%d per GPU in pieces of "
                                    "%d), 1%% subst errors, k=%d, -s %dM sharded by hash over %d GPU(s), -b %d, -q %d; FASTQ resident "
                                    "in HBM; step = clear + per piece (index + extract by owner%s + bucket) + one pass over the table "
                                    "per batch of pieces" % (genome_bases // 1_000_000, reads_rank * world, READ_LEN, reads_rank,
                                                             args.piece_reads, K, total_size // 1_000_000, world, BITS, MIN_QUAL,
                                                             " + all-to-all exchange" if grouped else "")),
                       "insertions_per_step": int(per_step), "distinct_per_gpu": int(st["distinct"]),
                       "slots_per_gpu": int(st["slots"]), "pieces_per_gpu": len(pieces),
                       "window_passes_per_step": int(st["window_passes"]),
                       "bucket_overflow_retries": int(getattr(b, "retries", 0)),
                       "exchange_rounds_max": int(getattr(b, "rounds_max", 0)), "parallelism": "hash-sharded x%d" % world},
            "roofline": roofline_block(value / world, ktimes, st, args.steps, single, per_step / world, traffic,
                                       shard_bits=world.bit_length() - 1),
        }
        if check is not None:
            out["content_check"] = check
        if grouped:
            out["exchange_message_cap"] = _max_message_items() * 8
            if big is not None:
                out["large_message_ok"] = big["mismatched_words_all_ranks"] == 0
                out["large_message_check"] = big
        if world == 1 and not args.no_cpu_baseline:
            if fq is None:
                fq = pieces[0][:min(piece_reads[0], 2_000_000) * (2 * READ_LEN + 16)].cpu().numpy()
            # a table the sample's distinct mers fill to about a half, like the real one
            out["cpu_baseline"] = cpu_baseline(fq, piece_reads[0], C1_SIZE if workload == "c1" else 1 << 28)
    ok = True
    if rank == 0 and check is not None and not check["ok"]:
        ok = False

    # N = 1: the configs[1] figures beside the configs[2] line (same process, same box)
    c1 = None
    if world == 1 and not grouped and workload == "c2" and not args.no_c1 and not args.reads:
        del b, pieces
        torch.cuda.empty_cache()
        p1, n1, fq1 = make_c1(0, 1, dev, C1_READS)
        b1 = ShardedBuilder(K, BITS, C1_SIZE, MIN_QUAL, 0, 1, local, own_stream=True)
        b1.h.set_mode(args.mode, args.level1_bits)
        dt1, kt1, st1 = measure(b1, p1, n1, max(args.steps, 10), max(args.warmup, 2), fence)
        steps1 = max(args.steps, 10)
        v1 = st1["instances"] * steps1 / dt1
        c1 = {"workload": "BASELINE configs[1]: synthetic 5 Mbp genome, 2000000 x 100 bp reads, k=24, -s 50M, -b 7, -q 38 "
                          "(the headline of rounds 1 and 2)",
              "value": v1, "unit": "k-mers/s", "ms_per_step": dt1 / steps1 * 1e3, "steps": steps1,
              "insertions_per_step": int(st1["instances"]),
              "roofline": roofline_block(v1, kt1, st1, steps1, True, float(st1["instances"]), pmc_traffic("whole_step_hbm_bytes"))}
        # measured random-access roofline: uniformly random 64-bit atomicCAS / load+CAS /
        # 8-byte loads over a table of this config's size, same number of updates as one step
        import ctypes as C
        from quorum_amd._lib import lib, check as qcheck
        lsize = int(st1["slots"]).bit_length() - 1
        ra = {}
        for name, mode in (("cas", 0), ("load_then_cas", 2), ("load8", 1)):
            ms = C.c_double()
            qcheck(lib.qh_bench_random_access(local, lsize, int(st1["instances"]), mode, 3, C.byref(ms)))
            ra[name + "_per_s"] = st1["instances"] / (ms.value * 1e-3)
        # not a roofline fraction: how the partitioned build's whole-step rate compares with what one global
        # atomic per k-mer could reach at best
        ra["step_kmers_per_s"] = v1
        ra["step_vs_random_cas"] = v1 / ra["cas_per_s"]
        c1["random_access_roofline"] = ra
        out["configs1"] = c1
    if rank == 0:
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit("bench.py: the content check failed (see content_check in the line above)")


if __name__ == "__main__":
    main()
