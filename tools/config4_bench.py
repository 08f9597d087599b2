#!/usr/bin/env python3
"""BASELINE configs[4] measured: 150 bp reads with 3 % substitution errors and low-quality tails, k = 31, adapters
(tests/golden/adapter.fa = the reference's data/adapter.fa) spliced into the 3' end of 1 % of the reads; database
build, then the corrector with the contaminant set (--trim-contaminant), both on text resident in HBM.

    python tools/config4_bench.py [--reads N] [--genome BASES] [--steps K] [--cpu-sample READS]
prints one JSON line: build rate (k-mers/s), corrector rate (bases/s), kernel times, and -- unless --cpu-sample 0 --
whether the GPU's output for the first READS reads equals the oracle's (database of ALL reads on both sides)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

K, BITS, MIN_QUAL, L = 31, 7, 38, 150
ADAPTERS = os.path.join(ROOT, "tests", "golden", "adapter.fa")


def splice_adapters(fq, rate, seed):
    """adapter prefixes over the 3' end of `rate` of the reads (record = 2 * L + 16 bytes, fixed)"""
    from quorum_amd import synth
    rec = synth.record_bytes(L)
    n = fq.size // rec
    assert n * rec == fq.size
    lines = [l.strip() for l in open(ADAPTERS, "rb").read().split(b"\n") if l and not l.startswith(b">")]
    ads = [np.frombuffer(a, dtype=np.uint8) for a in lines if len(a) >= K]
    rng = np.random.default_rng(seed)
    view = fq.reshape(n, rec)
    seq0 = int(np.flatnonzero(view[0] == 10)[0]) + 1          # the sequence line starts after the header's newline
    assert (view[:, seq0 - 1] == 10).all(), "headers of one width"
    for i in np.flatnonzero(rng.random(n) < rate):
        a = ads[int(rng.integers(0, len(ads)))]
        m = min(a.size, 60)
        view[i, seq0 + L - m:seq0 + L] = a[:m]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1_300_000)
    ap.add_argument("--genome", type=int, default=5_000_000)
    ap.add_argument("--size", type=int, default=400_000_000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=20_000)
    ap.add_argument("--err", type=float, default=0.03, help="substitution rate (parts of the configuration, for breakdowns)")
    ap.add_argument("--no-contaminant", action="store_true")
    ap.add_argument("--no-tails", action="store_true")
    a = ap.parse_args()

    import quorum_amd as qa
    from quorum_amd import synth

    g = synth.make_genome(a.genome, seed=51)
    fq = synth.make_fastq(g, a.reads, L, err=a.err, seed=52, tails=not a.no_tails)
    if not a.no_contaminant:
        splice_adapters(fq, 0.01, 55)
    chunk = torch.from_numpy(fq).cuda()
    torch.cuda.synchronize()

    h = qa.HashWithQuality(a.size, 2 * K, BITS, min_qual=MIN_QUAL)
    h.enable_timing(True)
    build_ms = []
    for i in range(a.steps + 1):
        h.reset()
        h.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h.add_fastq_device(chunk.data_ptr(), chunk.numel())
        h.sync()
        if i:
            build_ms.append((time.perf_counter() - t0) * 1e3)
    st = h.done()
    ktimes = [h.kernel_time(i) for i in range(4)]
    q = h.as_query()
    gc = None if a.no_contaminant else qa.contaminant_database(ADAPTERS, K)
    ec = qa.ErrorCorrector(q, contaminant=gc, trim_contaminant=True)
    ec.correct_device(chunk.data_ptr(), chunk.numel())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kt = []
    for _ in range(a.steps):
        ec.correct_device(chunk.data_ptr(), chunk.numel())
        kt.append(ec.kernel_time())
    dt = (time.perf_counter() - t0) / a.steps
    fa, lg, cst = ec.fetch()
    bms = sum(build_ms) / len(build_ms)
    out = {
        "config": {"workload": "BASELINE configs[4] shape: %d x %d bp reads of a %d Mbp genome, 3%% substitution errors, "
                               "low-quality tails, adapters in 1%% of the reads, k=%d, -s %dM -b %d -q %d, contaminant set "
                               "adapter.fa with --trim-contaminant; text resident in HBM"
                               % (a.reads, L, a.genome // 1_000_000, K, a.size // 1_000_000, BITS, MIN_QUAL)},
        "build": {"ms": bms, "kmers_per_s": st["instances"] / (bms * 1e-3), "instances": int(st["instances"]),
                  "distinct": int(st["distinct"]), "slots": int(st["slots"]), "grows": int(st["grows"]),
                  "kernel_ms": [ms / n if n else 0.0 for ms, n in ktimes]},
        "correct": {"ms": dt * 1e3, "bases_per_s": cst["bases_in"] / dt, "kernel_ms": sum(x for x, _ in kt) / len(kt),
                    "cutoff": int(ec.cutoff), "stats": cst, "fa_bytes": len(fa), "log_bytes": len(lg)},
    }
    if a.cpu_sample:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle as po
        n = min(a.cpu_sample, a.reads)
        odb = po.OracleDB(a.size, K, BITS)
        t0 = time.perf_counter()
        ni = odb.add_fastq_ptr(fq.ctypes.data, fq.size, MIN_QUAL, min(16, os.cpu_count() or 1))
        t_build = time.perf_counter() - t0
        oc = po.OracleDB(1 << 16, K, 1)
        if not a.no_contaminant:
            oc.add_fasta(open(ADAPTERS, "rb").read())
        sample = fq[:n * synth.record_bytes(L)].tobytes()
        t0 = time.perf_counter()
        ofa, olg, on = odb.correct_fastq(sample, po.ec_params(ec.cutoff, trim_contaminant=True),
                                          contaminant=None if a.no_contaminant else oc)
        t_corr = time.perf_counter() - t0
        out["oracle"] = {"build_kmers_per_s": ni / t_build, "build_threads": min(16, os.cpu_count() or 1),
                         "correct_bases_per_s_one_thread": n * L / t_corr, "sample_reads": n,
                         "instances_equal": bool(ni == st["instances"]),
                         "cutoff_equal": bool(odb.poisson_cutoff(0.01 / 3, 1e-6 / 0.01) == ec.cutoff),
                         "gpu_text_matches_oracle_on_sample": bool(fa.startswith(ofa) and lg.startswith(olg))}
        odb.close()
        oc.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
