#!/usr/bin/env python3
"""A database file far beyond 4 GiB through the tools: quorum_create_database -s 4G (2^32 slots: 34 GB of keys + 4 GB of
values), read back by DatabaseQuery and by quorum_error_correct_reads, and compared with a build of the same reads made in
this process (statistics, a million sampled lookups, a slice of the iteration).
    python tools/big_file_check.py [--reads N] [--size 4G] [--dir /tmp]"""
import argparse, json, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=6_000_000)
ap.add_argument("--genome", type=int, default=50_000_000)
ap.add_argument("--size", default="4G")
ap.add_argument("--dir", default="/tmp")
a = ap.parse_args()
import quorum_amd as qa
from quorum_amd import synth

K, BITS = 24, 7
g = synth.make_genome(a.genome)
fq = synth.make_fastq(g, a.reads, 100)
path, out = os.path.join(a.dir, "big_check.fastq"), os.path.join(a.dir, "big_check.jf")
fq.tofile(path)
bindir = os.path.join(ROOT, "quorum_amd", "bin")
res = {}
t0 = time.time()
r = subprocess.run([os.path.join(bindir, "quorum_create_database"), "-s", a.size, "-m", str(K), "-b", str(BITS), "-q", "38", "-o", out, path],
                   capture_output=True, text=True, env=dict(os.environ, QUORUM_HIP_VERBOSE="1"))
res["create_s"] = time.time() - t0
assert r.returncode == 0, r.stderr
res["create_stderr"] = r.stderr.strip()
res["file_bytes"] = os.path.getsize(out)
t0 = time.time()
q = qa.DatabaseQuery(out)
res["open_s"] = time.time() - t0
size = int(a.size[:-1]) * 10 ** 9 if a.size.endswith("G") else int(a.size)
h = qa.HashWithQuality(size, 2 * K, BITS, min_qual=38)
h.add_fastq(fq.tobytes())
st = h.done()
q2 = h.as_query()
s1, s2 = q.stats(), q2.stats()
res["stats_file"], res["stats_mem"] = s1, s2
assert s1 == s2, (s1, s2)
rng = np.random.default_rng(1)
# mers that are in the table (taken from reads) and random ones
starts = rng.integers(0, a.genome - K, size=500_000)
codes = g[starts[:, None] + np.arange(K)[None, :]].astype(np.uint64)
mers = (codes << (2 * (K - 1 - np.arange(K, dtype=np.uint64)))[None, :]).sum(axis=1).astype(np.uint64)
rnd = rng.integers(0, 1 << (2 * K), size=500_000, dtype=np.uint64)
allm = np.concatenate([mers, rnd])
v1, v2 = q.lookup(allm), q2.lookup(allm)
assert (v1 == v2).all()
res["lookups_equal"] = True
res["hits_among_genome_mers"] = int((v1[:500_000] != 0).sum())
t0 = time.time()
r = subprocess.run([os.path.join(bindir, "quorum_error_correct_reads"), "-v", out, "-o", os.path.join(a.dir, "big_check_ec"), path],
                   capture_output=True, text=True)
res["correct_s"] = time.time() - t0
assert r.returncode == 0, r.stderr
res["correct_stderr"] = r.stderr.strip().splitlines()[-4:]
# the same chunk corrected in this process from the in-memory table: the first records must agree
ec = qa.ErrorCorrector(q2)
n = 200_000 * synth.record_bytes(100)
fa, lg, cst, used = ec.correct_fastq(fq[:n].tobytes())
with open(os.path.join(a.dir, "big_check_ec.fa"), "rb") as f:
    head = f.read(len(fa))
res["corrector_text_equal_on_first_200k"] = bool(head == fa)
assert head == fa
print(json.dumps(res))
for f in ("big_check.fastq", "big_check.jf", "big_check_ec.fa", "big_check_ec.log"):
    try:
        os.remove(os.path.join(a.dir, f))
    except OSError:
        pass
