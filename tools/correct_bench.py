#!/usr/bin/env python3
"""Throughput of the GPU read corrector on BASELINE configs[1]: 2M x 100 bp reads of a
5 Mbp genome, 1 % errors, k=24. The database is built from the same reads on the GPU; a step is
one pass of qh_corrector_correct_device over the FASTQ text resident in HBM (record index +
k_correct_reads + output text), timed with the stream synchronised on both sides.

Unit of work = one input base. Algorithmic bytes per base: 4 table slots x 8 B (one
get_best_alternatives per extended base) + 1 B read + 1 B written = 34 B. The four slots share one
probe sequence (core hash), so the memory-side bound is ONE random 64-byte line per extended base:
the random line rate of HBM (qh_bench_random_access mode 1), not its streaming rate.

    python tools/correct_bench.py [--reads N] [--steps K] [--warmup W] [--cpu-sample READS]
prints one JSON line."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

K, BITS, MIN_QUAL, READ_LEN = 24, 7, 38, 100


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2_000_000)
    ap.add_argument("--genome", type=int, default=5_000_000)
    ap.add_argument("--size", type=int, default=50_000_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=100_000, help="reads the oracle corrects for the CPU baseline (0 = skip)")
    ap.add_argument("--err", type=float, default=0.01, help="substitution rate of the reads (0: the kernel's straight path only)")
    ap.add_argument("--db-err", type=float, default=-1, help="substitution rate of the reads the database is built from (default: the corrected reads themselves)")
    ap.add_argument("--fetch", action="store_true", help="also time the copy of both text streams to the host")
    args = ap.parse_args()

    import quorum_amd as qa
    from quorum_amd import synth
    from quorum_amd._lib import lib, check

    genome = synth.make_genome(args.genome, seed=42)
    fq = synth.make_fastq(genome, args.reads, READ_LEN, err=args.err, n_rate=0.0005 if args.err else 0.0, seed=42)
    chunk = torch.from_numpy(fq).cuda()
    torch.cuda.synchronize()
    h = qa.HashWithQuality(args.size, 2 * K, BITS, min_qual=MIN_QUAL)
    if args.db_err >= 0:
        dbchunk = torch.from_numpy(synth.make_fastq(genome, args.reads, READ_LEN, err=args.db_err, n_rate=0.0005, seed=42)).cuda()
        torch.cuda.synchronize()
        h.add_fastq_device(dbchunk.data_ptr(), dbchunk.numel())
    else:
        h.add_fastq_device(chunk.data_ptr(), chunk.numel())
    st_build = h.done()
    q = h.as_query()
    ec = qa.ErrorCorrector(q)

    def step():
        ec.correct_device(chunk.data_ptr(), chunk.numel())
        return ec.kernel_time()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kt = [step() for _ in range(args.steps)]
    dt = (time.perf_counter() - t0) / args.steps
    t1 = time.perf_counter()
    fa, lg, st = ec.fetch()
    t_fetch = time.perf_counter() - t1
    k_ms = sum(a for a, _ in kt) / len(kt)
    c_ms = sum(b for _, b in kt) / len(kt)

    lsize = int(st_build["slots"]).bit_length() - 1
    ms = C.c_double()
    n_probe = st["bases_in"]
    check(lib.qh_bench_random_access(0, lsize, int(n_probe), 1, 3, C.byref(ms)))
    load_rate = n_probe / (ms.value * 1e-3)

    bases = st["bases_in"]
    extended = bases - st["reads"] * (K + 1)          # bases that go through extend() when nothing is cut short
    out = {
        "metric": "bases/s through the read corrector", "value": bases / dt, "unit": "bases/s",
        "reads_per_s": st["reads"] / dt, "ms_per_step": dt * 1e3, "steps": args.steps, "warmup": args.warmup,
        "config": {"workload": "BASELINE configs[1] reads (%d x %d bp, %d Mbp genome, 1%% errors), k=%d, database of the "
                               "same reads (-s %dM -b %d -q %d), default corrector switches, cutoff %d; FASTQ resident in HBM"
                               % (args.reads, READ_LEN, args.genome // 1_000_000, K, args.size // 1_000_000, BITS, MIN_QUAL,
                                  ec.cutoff)},
        "stats": st, "fa_bytes": len(fa), "log_bytes": len(lg),
        "kernel": {"name": "k_correct_reads", "avg_ms": k_ms, "chunk_ms_by_events": c_ms},
        "roofline": {"bound": "hbm (random 8-byte loads)", "alg_bytes_per_base": 34.0,
                     "achieved_GBs": bases * 34.0 / (k_ms * 1e-3) / 1e9, "peak_GBs": 8000.0,
                     "frac": bases * 34.0 / (k_ms * 1e-3) / 1e9 / 8000.0,
                     "sibling_walks_per_s": extended / (k_ms * 1e-3),
                     "measured_random_line_rate_per_s": load_rate,
                     "frac_of_random_line_rate": extended / (k_ms * 1e-3) / load_rate},
        "fetch_seconds": t_fetch,
    }
    if args.cpu_sample:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle as po
        n = min(args.cpu_sample, args.reads)
        nbytes = n * synth.record_bytes(READ_LEN)
        odb = po.OracleDB(args.size, K, BITS)
        odb.add_fastq_ptr(fq.ctypes.data, fq.size, MIN_QUAL, os.cpu_count() or 1)
        sample = fq[:nbytes].tobytes()
        t0 = time.perf_counter()
        ofa, olg, on = odb.correct_fastq(sample, po.ec_params(ec.cutoff))
        cdt = time.perf_counter() - t0
        # parity on the sample while we are here: the GPU text starts with the oracle's
        out["cpu_baseline"] = {"value": n * READ_LEN / cdt, "unit": "bases/s", "cores": 1, "kind": "port",
                               "sample": "first %d reads, one thread of oracle/error_correct_oracle.c (%.2f s)" % (n, cdt),
                               "gpu_text_matches_oracle_on_sample": bool(fa.startswith(ofa) and lg.startswith(olg))}
        odb.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
