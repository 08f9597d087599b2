"""End-to-end: the drop-in quorum_create_database binary writes a database that the
query tools and database_query read back with the oracle's content."""
import os
import subprocess

import numpy as np
import pytest

import pyoracle as po
import quorum_amd as qa
from quorum_amd import synth
from helpers import sorted_pairs, mer_from_str

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "quorum_amd", "bin")


def test_create_database_cli_end_to_end(tmp_path):
    k, bits = 24, 7
    g = synth.make_genome(100000)
    fq1 = synth.make_fastq(g, 30000, 100, tails=True).tobytes()
    fq2 = synth.make_fastq(g, 10000, 150, err=0.03, first_id=30000).tobytes()
    # second file in multi-line form: exercises the host-parser fallback inside the CLI
    recs = fq2.split(b"\n")
    multi = []
    for i in range(0, len(recs) - 1, 4):
        s, q = recs[i + 1], recs[i + 3]
        multi += [recs[i], s[:70], s[70:], b"+", q[:70], q[70:]]
    fq2m = b"\n".join(multi) + b"\n"
    (tmp_path / "a.fastq").write_bytes(fq1)
    (tmp_path / "b.fastq").write_bytes(fq2m)
    out = str(tmp_path / "db.jf")
    env = dict(os.environ, QUORUM_HIP_CHUNK_MB="2", QUORUM_HIP_VERBOSE="1")  # many chunks
    r = subprocess.run([os.path.join(BIN, "quorum_create_database"), "-s", "200k", "-m", str(k), "-b", str(bits),
                        "-q", "38", "-t", "4", "-o", out, str(tmp_path / "a.fastq"), str(tmp_path / "b.fastq")],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    db = po.OracleDB(1 << 23, k, bits)
    n = db.add_fastq(fq1, 38, threads=8) + db.add_fastq(fq2, 38, threads=8)
    assert ("k-mers %d " % n) in r.stderr
    om, ov = db.sorted_dump()
    q = qa.DatabaseQuery(out)
    m, v = sorted_pairs(*q.dump())
    assert m.size == om.size and (m == om).all() and (v == ov).all()
    # query_mer_database: "<arg>:<canonical> val:<count> qual:<hq>" (query_mer_database.cc:14-21)
    s = qa.mer_to_str(int(om[5]), k)
    rc = qa.mer_to_str(qa.revcomp(int(om[5]), k), k)
    r = subprocess.run([os.path.join(BIN, "query_mer_database"), out, rc, "A" * k], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split("\n")
    assert lines[0] == str(k)
    assert lines[1] == "%s:%s val:%d qual:%d" % (rc, s, int(ov[5]) >> 1, int(ov[5]) & 1)
    poly = int(db.lookup([0])[0])
    assert lines[2] == "%s:%s val:%d qual:%d" % ("A" * k, "A" * k, poly >> 1, poly & 1)
    # histo_mer_database: "count lowq highq" (histo_mer_database.cc:23-26)
    r = subprocess.run([os.path.join(BIN, "histo_mer_database"), out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    h = db.histo()
    want = ["%d %d %d" % (i, h[i, 0], h[i, 1]) for i in range(1001) if h[i, 0] or h[i, 1]]
    assert r.stdout.strip().split("\n") == want
    q.close()


def test_cli_min_qual_char_and_missing_file(tmp_path):
    fq = b"@a\nACGTACGTAC\n+\nIIII#IIIII\n"
    (tmp_path / "r.fq").write_bytes(fq)
    out = str(tmp_path / "o.db")
    r = subprocess.run([os.path.join(BIN, "quorum_create_database"), "-s", "1k", "-m", "4", "-b", "7", "-Q", "&",
                        "-o", out, str(tmp_path / "r.fq")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    db = po.OracleDB(1024, 4, 7)
    db.add_fastq(fq, ord("&"))
    q = qa.DatabaseQuery(out)
    m, v = sorted_pairs(*q.dump())
    om, ov = db.sorted_dump()
    assert (m == om).all() and (v == ov).all()
    q.close()
    r = subprocess.run([os.path.join(BIN, "quorum_create_database"), "-s", "1k", "-m", "4", "-b", "7", "-q", "38",
                        "-o", out, str(tmp_path / "missing.fq")], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
