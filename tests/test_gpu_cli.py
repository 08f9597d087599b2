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
ADAPTERS = os.path.join(ROOT, "tests", "golden", "adapter.fa")


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


@pytest.mark.parametrize("devices", ["0,0", "0,0,0,0"])
def test_create_database_cli_sharded_one_process(tmp_path, devices):
    """quorum_create_database --devices a,b,...: the hash-sharded build from ONE process (qh_multi_*: a host thread per
    shard, owner-bucketed extraction, all-to-all by peer copies, shards serialised into one file). The one-GPU box names
    device 0 for every shard; with distinct devices the same code exchanges over RCCL. First file strict 4-line FASTQ in
    many small rounds, second file multi-line (the host-parser route on every shard)."""
    k, bits = 24, 7
    g = synth.make_genome(100000)
    fq1 = synth.make_fastq(g, 30000, 100, tails=True).tobytes()
    fq2 = synth.make_fastq(g, 6000, 150, err=0.03, first_id=30000).tobytes()
    recs = fq2.split(b"\n")
    multi = []
    for i in range(0, len(recs) - 1, 4):
        s, q = recs[i + 1], recs[i + 3]
        multi += [recs[i], s[:70], s[70:], b"+", q[:70], q[70:]]
    (tmp_path / "a.fastq").write_bytes(fq1)
    (tmp_path / "b.fastq").write_bytes(b"\n".join(multi) + b"\n")
    out = str(tmp_path / "db.jf")
    env = dict(os.environ, QUORUM_HIP_CHUNK_MB="1", QUORUM_HIP_VERBOSE="1")
    r = subprocess.run([os.path.join(BIN, "quorum_create_database"), "-s", "400k", "-m", str(k), "-b", str(bits),
                        "-q", "38", "--devices", devices, "-o", out, str(tmp_path / "a.fastq"), str(tmp_path / "b.fastq")],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    db = po.OracleDB(1 << 23, k, bits)
    n = db.add_fastq(fq1, 38, threads=8) + db.add_fastq(fq2, 38, threads=8)
    assert ("k-mers %d " % n) in r.stderr and ("reads %d " % 36000) in r.stderr
    assert "exchange peer copies" in r.stderr and "host-parsed chunks 0" not in r.stderr
    om, ov = db.sorted_dump()
    q = qa.DatabaseQuery(out)
    m, v = sorted_pairs(*q.dump())
    assert m.size == om.size and (m == om).all() and (v == ov).all()
    assert (q.lookup(om[::11]) == ov[::11]).all()
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


def test_quorum_pipeline_cli_end_to_end(tmp_path):
    """what the quorum driver runs (src/quorum.in:155-171): quorum_create_database, then
    quorum_error_correct_reads on the same files with the database; .fa and .log against the oracle"""
    k, bits = 24, 7
    g = synth.make_genome(40000, seed=21)
    fq1 = synth.make_fastq(g, 9000, 100, seed=22, tails=True).tobytes()
    fq2 = synth.make_fastq(g, 5000, 120, err=0.02, seed=23, first_id=9000).tobytes()
    recs = fq2.split(b"\n")
    multi = []
    for i in range(0, len(recs) - 1, 4):  # second file in multi-line form: the host parser inside the CLI
        s, q = recs[i + 1], recs[i + 3]
        multi += [recs[i], s[:70], s[70:], b"+", q[:70], q[70:]]
    fq2m = b"\n".join(multi) + b"\n"
    (tmp_path / "a.fastq").write_bytes(fq1)
    (tmp_path / "b.fastq").write_bytes(fq2m)
    dbf = str(tmp_path / "q_mer_database.jf")
    env = dict(os.environ, QUORUM_HIP_CHUNK_MB="1")  # several chunks per file, records cut at chunk ends
    r = subprocess.run([os.path.join(BIN, "quorum_create_database"), "-s", "1M", "-m", str(k), "-b", str(bits),
                        "-q", "38", "-o", dbf, str(tmp_path / "a.fastq"), str(tmp_path / "b.fastq")],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    prefix = str(tmp_path / "q")
    r = subprocess.run([os.path.join(BIN, "quorum_error_correct_reads"), "-t", "4", "--contaminant", ADAPTERS,
                        "--trim-contaminant", "-d", "-v", dbf, "-o", prefix, str(tmp_path / "a.fastq"),
                        str(tmp_path / "b.fastq")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert "Using cutoff of" in r.stderr
    db = po.OracleDB(1 << 20, k, bits)
    db.add_fastq(fq1, 38)
    db.add_fastq(fq2, 38)
    oc = po.OracleDB(1 << 16, k, 1)
    oc.add_fasta(open(ADAPTERS, "rb").read())
    cut = db.poisson_cutoff(0.01 / 3, 1e-6 / 0.01)
    assert ("Using cutoff of %d" % cut) in r.stderr
    par = po.ec_params(cut, trim_contaminant=True, no_discard=True)
    fa1, lg1, n1 = db.correct_fastq(fq1, par, contaminant=oc)
    fa2, lg2, n2 = db.correct_fastq(fq2m, par, contaminant=oc)
    assert open(prefix + ".fa", "rb").read() == fa1 + fa2
    assert open(prefix + ".log", "rb").read() == lg1 + lg2
    assert ("reads %d " % (n1 + n2)) in r.stderr
    # without -o the streams go to stdout / stderr (error_correct_reads.cc:160-161,723)
    r = subprocess.run([os.path.join(BIN, "quorum_error_correct_reads"), "-p", str(cut), "-a", "200", dbf,
                        str(tmp_path / "a.fastq")], capture_output=True, timeout=300, env=env)
    assert r.returncode == 0
    fa, lg, n = db.correct_fastq(fq1, po.ec_params(cut, anchor=200))
    assert r.stdout == fa and r.stderr == lg and lg.count(b"Skipped") == 9000
    # argument rules of main (:680-683) and of the switch parser
    for argv, msg in ((["-Q", "ab", dbf, "x"], "must be one ASCII character"),
                      (["-q", "200", dbf, "x"], "must be in the range 0-127"),
                      (["-q", "40", "-Q", "I", dbf, "x"], "mutually exclusive"),
                      ([dbf], "Requires at least 2 arguments"),
                      (["--gzip", dbf, "x"], "not supported")):
        r = subprocess.run([os.path.join(BIN, "quorum_error_correct_reads")] + argv, capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and msg in r.stderr, (argv, r.stderr)
    r = subprocess.run([os.path.join(BIN, "quorum_error_correct_reads"), str(tmp_path / "missing.jf"), "x"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 1


def test_paired_end_pipeline(tmp_path):
    """quorum's paired-end plumbing (src/quorum.in:172-231): merge_mate_pairs f1 f2 | error_correct_reads -d db
    /dev/fd/0 2> log | split_mate_pairs prefix -- mates stay in step because -d keeps every read"""
    k, bits = 24, 7
    g = synth.make_genome(30000, seed=41)
    fq1 = synth.make_fastq(g, 6000, 100, seed=42, tails=True).tobytes()
    fq2 = synth.make_fastq(g, 6000, 100, seed=43, tails=True, first_id=6000).tobytes()
    (tmp_path / "r1.fastq").write_bytes(fq1)
    (tmp_path / "r2.fastq").write_bytes(fq2)
    dbf = str(tmp_path / "db.jf")
    r = subprocess.run([os.path.join(BIN, "quorum_create_database"), "-s", "1M", "-m", str(k), "-b", str(bits), "-q", "38",
                        "-o", dbf, str(tmp_path / "r1.fastq"), str(tmp_path / "r2.fastq")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    merge = subprocess.Popen([os.path.join(BIN, "merge_mate_pairs"), str(tmp_path / "r1.fastq"), str(tmp_path / "r2.fastq")],
                             stdout=subprocess.PIPE)
    with open(tmp_path / "p.log", "wb") as lg:
        correct = subprocess.Popen([os.path.join(BIN, "quorum_error_correct_reads"), "-d", "-a", "25", dbf, "/dev/fd/0"],
                                   stdin=merge.stdout, stdout=subprocess.PIPE, stderr=lg,
                                   env=dict(os.environ, QUORUM_HIP_CHUNK_MB="1"))
        merge.stdout.close()
        split = subprocess.Popen([os.path.join(BIN, "split_mate_pairs"), str(tmp_path / "p")], stdin=correct.stdout)
        correct.stdout.close()
        assert split.wait(timeout=300) == 0 and correct.wait(timeout=300) == 0 and merge.wait(timeout=300) == 0
    db = po.OracleDB(1 << 20, k, bits)
    db.add_fastq(fq1, 38)
    db.add_fastq(fq2, 38)
    cut = db.poisson_cutoff(0.01 / 3, 1e-6 / 0.01)
    l1, l2 = fq1.split(b"\n"), fq2.split(b"\n")
    inter = b"".join(b"\n".join(l1[i:i + 4]) + b"\n" + b"\n".join(l2[i:i + 4]) + b"\n" for i in range(0, len(l1) - 1, 4))
    fa, lg, n = db.correct_fastq(inter, po.ec_params(cut, anchor=25, no_discard=True))
    assert n == 12000 and lg.count(b"Skipped") > 10          # some reads are discarded: a single N keeps the pairing
    out = fa.split(b"\n")
    want1 = b"".join(out[i] + b"\n" + out[i + 1] + b"\n" for i in range(0, len(out) - 1, 4))
    want2 = b"".join(out[i + 2] + b"\n" + out[i + 3] + b"\n" for i in range(0, len(out) - 1, 4))
    assert (tmp_path / "p_1.fa").read_bytes() == want1
    assert (tmp_path / "p_2.fa").read_bytes() == want2
    assert (tmp_path / "p.log").read_bytes() == lg
    assert (tmp_path / "p_1.fa").read_bytes().count(b">") == 6000


def test_quorum_driver(tmp_path):
    """the driver (src/quorum.in): min-q-char autodetection, database, correction; -P keeps mates in two files"""
    import sys
    drv = [sys.executable, os.path.join(ROOT, "quorum_amd", "quorum_driver.py")]
    k = 24
    g = synth.make_genome(30000, seed=61)
    fq1 = synth.make_fastq(g, 5000, 100, seed=62, tails=True).tobytes()
    fq2 = synth.make_fastq(g, 5000, 100, seed=63, tails=True, first_id=5000).tobytes()
    (tmp_path / "r1.fastq").write_bytes(fq1)
    (tmp_path / "r2.fastq").write_bytes(fq2)
    prefix = str(tmp_path / "qc")
    r = subprocess.run(drv + ["-s", "1M", "-p", prefix, "--debug", str(tmp_path / "r1.fastq"), str(tmp_path / "r2.fastq")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "-q 38 -b 7" in r.stderr                      # detected 33 ('#' = 35 -> 33), + min-quality 5 (quorum.in:146,156)
    db = po.OracleDB(1 << 20, k, 7)
    db.add_fastq(fq1, 38)
    db.add_fastq(fq2, 38)
    cut = db.poisson_cutoff(0.01 / 3, 1e-6 / 0.01)
    fa1, lg1, _ = db.correct_fastq(fq1, po.ec_params(cut))
    fa2, lg2, _ = db.correct_fastq(fq2, po.ec_params(cut))
    assert open(prefix + ".fa", "rb").read() == fa1 + fa2 and open(prefix + ".log", "rb").read() == lg1 + lg2
    assert os.path.exists(prefix + "_mer_database.jf")
    # paired files, window / error / homo-trim passed through, -d implied
    r = subprocess.run(drv + ["-s", "1M", "-p", prefix, "-P", "-w", "8", "-e", "2", "--homo-trim", "3",
                              str(tmp_path / "r1.fastq"), str(tmp_path / "r2.fastq")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    l1, l2 = fq1.split(b"\n"), fq2.split(b"\n")
    inter = b"".join(b"\n".join(l1[i:i + 4]) + b"\n" + b"\n".join(l2[i:i + 4]) + b"\n" for i in range(0, len(l1) - 1, 4))
    fa, lg, n = db.correct_fastq(inter, po.ec_params(cut, window=8, error=2, homo_trim=3, no_discard=True))
    out = fa.split(b"\n")
    assert open(prefix + "_1.fa", "rb").read() == b"".join(out[i] + b"\n" + out[i + 1] + b"\n" for i in range(0, len(out) - 1, 4))
    assert open(prefix + "_2.fa", "rb").read() == b"".join(out[i + 2] + b"\n" + out[i + 3] + b"\n" for i in range(0, len(out) - 1, 4))
    assert open(prefix + ".log", "rb").read() == lg
    # argument rules (:92-99)
    r = subprocess.run(drv + ["-s", "12x", "f"], capture_output=True, text=True)
    assert r.returncode == 1 and "Invalid size '12x'" in r.stderr
    r = subprocess.run(drv + ["-s", "1M"], capture_output=True, text=True)
    assert r.returncode == 1 and "No sequence files" in r.stderr
    (tmp_path / "odd.fastq").write_bytes(b"@r\nACGT\n+\nKKKK\n")   # minimum quality 'K' = 75: not a known offset
    r = subprocess.run(drv + ["-s", "1M", str(tmp_path / "odd.fastq")], capture_output=True, text=True)
    assert r.returncode == 1 and "unusual minimum quality char" in r.stderr


def test_sharded_cli_strict_input_never_takes_the_host_parser(tmp_path):
    """the parts a chunk is cut into for the shards begin at record starts (found from a few lines after each cut point):
    with strict 4-line input, quality lines starting with '@' and '+' included, no part is rejected by the GPU indexer"""
    k, bits = 24, 7
    g = synth.make_genome(60000)
    arr = synth.make_fastq(g, 20000, 100, tails=True).reshape(20000, synth.record_bytes(100))
    arr[::7, 15 + 100] = ord("@")   # the first quality character of every seventh record
    fq = arr.tobytes()
    quals = fq.split(b"\n")[3::4]
    assert sum(q.startswith(b"@") for q in quals) > 2000 and any(q.startswith(b"+") for q in quals)
    (tmp_path / "a.fastq").write_bytes(fq)
    out = str(tmp_path / "db.jf")
    env = dict(os.environ, QUORUM_HIP_CHUNK_MB="1", QUORUM_HIP_VERBOSE="1")
    r = subprocess.run([os.path.join(BIN, "quorum_create_database"), "-s", "400k", "-m", str(k), "-b", str(bits),
                        "-q", "38", "--devices", "0,0,0,0", "-o", out, str(tmp_path / "a.fastq")],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert "host-parsed chunks 0 " in r.stderr + " ", r.stderr
    db = po.OracleDB(1 << 22, k, bits)
    db.add_fastq(fq, 38, threads=8)
    om, ov = db.sorted_dump()
    q = qa.DatabaseQuery(out)
    m, v = sorted_pairs(*q.dump())
    assert m.size == om.size and (m == om).all() and (v == ov).all()
    q.close()


def test_corrector_piped_input_equals_file_input(tmp_path):
    """quorum_error_correct_reads reading a pipe (a second thread reads ahead into the next chunk's buffer) writes what it
    writes for the same bytes in a file: many small chunks with records cut at their ends, multi-line records, a read
    far longer than a buffer's spare room (the loop hands over to the plain one), and a truncated stream's error"""
    k, bits = 24, 7
    g = synth.make_genome(30000, seed=71)
    fq = synth.make_fastq(g, 14000, 100, seed=72, tails=True).tobytes()
    lines = fq.split(b"\n")
    multi = []
    for i in range(0, 4 * 3000, 4):          # the first 3000 records in multi-line form
        multi += [lines[i], lines[i + 1][:33], lines[i + 1][33:], b"+", lines[i + 3][:50], lines[i + 3][50:]]
    long_seq = bytes(synth._BASES[synth.make_genome(700000, seed=73)].tolist())
    long_rec = b"@long\n" + long_seq + b"\n+\n" + b"I" * len(long_seq) + b"\n"
    half = 7000 * synth.record_bytes(100)
    texts = {"strict": fq, "multi": b"\n".join(multi) + b"\n" + fq[half:], "long": fq[:half] + long_rec + fq[half:]}
    (tmp_path / "db.fastq").write_bytes(fq)
    dbf = str(tmp_path / "db.jf")
    r = subprocess.run([os.path.join(BIN, "quorum_create_database"), "-s", "1M", "-m", str(k), "-b", str(bits), "-q", "38",
                        "-o", dbf, str(tmp_path / "db.fastq")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    ec = os.path.join(BIN, "quorum_error_correct_reads")
    env = dict(os.environ, QUORUM_HIP_CHUNK_MB="1")
    for name, text in texts.items():
        (tmp_path / "in.fastq").write_bytes(text)
        rf = subprocess.run([ec, "-d", "-p", "4", dbf, str(tmp_path / "in.fastq")], capture_output=True, timeout=300, env=env)
        assert rf.returncode == 0, rf.stderr[-500:]
        rp = subprocess.run([ec, "-d", "-p", "4", dbf, "/dev/fd/0"], input=text, capture_output=True, timeout=300, env=env)
        assert rp.returncode == 0, rp.stderr[-500:]
        assert rp.stdout == rf.stdout and rp.stderr == rf.stderr, name
        assert sum(l.startswith(b">") for l in rp.stdout.split(b"\n")) == {"strict": 14000, "multi": 10000, "long": 14001}[name]
        rs = subprocess.run([ec, "-d", "-p", "4", dbf, "/dev/fd/0"], input=text, capture_output=True, timeout=300,
                            env=dict(env, QUORUM_HIP_NO_READ_AHEAD="1"))
        assert rs.returncode == 0 and rs.stdout == rf.stdout and rs.stderr == rf.stderr, name
        # replicas (--devices a,b,...: here three on the one GPU): every chunk cut into parts at record boundaries, the
        # parts corrected side by side, the text in input order -- the same bytes
        rr = subprocess.run([ec, "-d", "-p", "4", "--devices", "0,0,0", dbf, str(tmp_path / "in.fastq")], capture_output=True,
                            timeout=300, env=env)
        assert rr.returncode == 0 and rr.stdout == rf.stdout and rr.stderr == rf.stderr, name
    cut = fq[:len(fq) - 57]
    rp = subprocess.run([ec, "-d", "-p", "4", "--devices", "0,0", dbf, "/dev/fd/0"], input=cut, capture_output=True, timeout=300, env=env)
    (tmp_path / "in.fastq").write_bytes(cut)
    rf = subprocess.run([ec, "-d", "-p", "4", dbf, str(tmp_path / "in.fastq")], capture_output=True, timeout=300, env=env)
    assert rp.returncode != 0 and rf.returncode != 0 and rp.stderr[-200:] == rf.stderr[-200:]
