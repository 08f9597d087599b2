"""--jf-layout end to end on the GPU: build -> file in the recalled Jellyfish layout -> qh_query_open -> same content.

PARITY UNPINNED for the layout itself (see tests/test_jf_layout.py): this is the self round trip the isolated module
is tested by, with the content checked against the oracle's database like every other serialisation test."""
import os
import subprocess

import numpy as np
import pytest

import pyoracle as po
import quorum_amd as qa
from quorum_amd import mer_database as md
from quorum_amd import synth
from helpers import sorted_pairs

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "quorum_amd", "bin")


@pytest.mark.parametrize("k,bits", [(24, 7), (31, 5)])
def test_cli_jf_layout_round_trip(tmp_path, k, bits):
    g = synth.make_genome(80000)
    fq = synth.make_fastq(g, 20000, 100, tails=True).tobytes()
    (tmp_path / "a.fastq").write_bytes(fq)
    out = str(tmp_path / "db.jf")
    r = subprocess.run([os.path.join(BIN, "quorum_create_database"), "-s", "300k", "-m", str(k), "-b", str(bits),
                        "-q", "38", "--jf-layout", "-o", out, str(tmp_path / "a.fastq")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    db = po.OracleDB(1 << 22, k, bits)
    db.add_fastq(fq, 38, threads=8)
    om, ov = db.sorted_dump()
    # the host-side reader
    k2, b2, _, m, v = md.jf_read_file(out)
    m, v = sorted_pairs(m, v)
    assert (k2, b2) == (k, bits) and m.size == om.size and (m == om).all() and (v == ov).all()
    # database_query over the same file: header, lookups, dump, and the tools on top
    q = qa.DatabaseQuery(out)
    m, v = sorted_pairs(*q.dump())
    assert m.size == om.size and (m == om).all() and (v == ov).all()
    rng = np.random.default_rng(3)
    absent = rng.integers(0, 1 << (2 * k), size=1000, dtype=np.uint64)
    probe = np.concatenate([om[::7], absent])
    assert (q.lookup(probe) == db.lookup(probe)).all()
    q.close()
    r = subprocess.run([os.path.join(BIN, "histo_mer_database"), out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    h = db.histo()
    want = ["%d %d %d" % (i, h[i, 0], h[i, 1]) for i in range(1001) if h[i, 0] or h[i, 1]]
    assert r.stdout.strip().split("\n") == want


def test_write_jf_from_handle_and_not_with_gpus(tmp_path):
    k, bits = 20, 7
    g = synth.make_genome(20000)
    fq = synth.make_fastq(g, 4000, 100).tobytes()
    h = qa.HashWithQuality(1 << 18, 2 * k, bits, min_qual=38)
    h.add_fastq(fq)
    h.done()
    hm, hv = sorted_pairs(*h.dump())
    h.write_jf(tmp_path / "x.jf", argv=["t"])
    _, _, size, m, v = md.jf_read_file(tmp_path / "x.jf")
    m, v = sorted_pairs(m, v)
    assert size == 1 << 18 and (m == hm).all() and (v == hv).all()
    (tmp_path / "a.fastq").write_bytes(fq)
    r = subprocess.run([os.path.join(BIN, "quorum_create_database"), "-s", "300k", "-m", "20", "-b", "7", "--jf-layout",
                        "--devices", "0,0", "-o", str(tmp_path / "y"), str(tmp_path / "a.fastq")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "--jf-layout" in r.stderr
