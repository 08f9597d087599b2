"""The CPU oracle against the reference's own known answers
(unit_tests/test_mer_database.cc:115-120), the hand-derived fixtures of
tests/golden/  and two independent models. CPU only."""
import os
import numpy as np
import pytest

import pyoracle as po
from helpers import (golden_cases, mer_from_str, revcomp, window_kmers, dict_model, dict_model_np,
                     split_fastq, random_fastq)


def seq_mers(codes, k):
    """all k-mers of a code array, NOT canonicalised (as test_mer_database.cc:21-25 inserts them)"""
    n = codes.size - k + 1
    m = np.zeros(n, dtype=np.uint64)
    for i in range(k):
        m = (m << np.uint64(2)) | codes[i:i + n].astype(np.uint64)
    return m


@pytest.mark.parametrize("param", [1, 10])
def test_reference_six_classes(param):
    """TEST_P(MerDatabase, WriteRead): hq2,hq1,lqhq,hqlq,lq1,lq2 -> (2,1),(1,1),(1,1),(1,1),(1,0),(2,0);
    param 1 makes the table grow several times (test_mer_database.cc:70,138)."""
    rng = np.random.default_rng(7)
    k, bits, n = 31, 4, 20000
    seqs = {name: rng.integers(0, 4, size=n) for name in ("hq2", "hq1", "lqhq", "hqlq", "lq1", "lq2")}
    mers = {name: seq_mers(c, k) for name, c in seqs.items()}
    db = po.OracleDB(param * n // 5, k, bits)
    plan = [("hq2", 1), ("hq2", 1), ("hq1", 1), ("lqhq", 0), ("lqhq", 1), ("hqlq", 1), ("hqlq", 0),
            ("lq1", 0), ("lq2", 0), ("lq2", 0)]
    for name, q in plan:
        db.add_mers(mers[name], np.full(mers[name].size, q, dtype=np.uint8))
    expect = {"hq2": (2, 1), "hq1": (1, 1), "lqhq": (1, 1), "hqlq": (1, 1), "lq1": (1, 0), "lq2": (2, 0)}
    seen = {}
    for name, (c, q) in expect.items():
        v = db.lookup(mers[name])
        assert (v >> np.uint64(1) == c).all() and (v & np.uint64(1) == q).all(), name
        for m, x in zip(mers[name].tolist(), v.tolist()):
            seen[m] = x
    # iteration: every stored entry is one of the inserted mers with the same value (:123-134)
    dm, dv = db.dump()
    assert dm.size == len(seen)
    for m, x in zip(dm.tolist(), dv.tolist()):
        assert seen[m] == x
    if param == 1:
        assert db.size() > param * n // 5 * 2  # it did grow


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_golden_known_answers(case):
    db = po.OracleDB(64, case["k"], case["bits"])
    db.add_fastq(case["data"], case["min_qual"])
    m, v = db.sorted_dump()
    got = {int(a): (int(b) >> 1, int(b) & 1) for a, b in zip(m, v)}
    want = {mer_from_str(s): tuple(cv) for s, cv in case["expected"].items()}
    assert got == want
    # anything else is absent -> (0,0) (mer_database.hpp:285)
    k = case["k"]
    absent = [x for x in range(min(4 ** k, 300)) if x not in want]
    assert (db.lookup(np.array(absent, dtype=np.uint64)) == 0).all()


def test_extraction_matches_window_model():
    rng = np.random.default_rng(3)
    data = random_fastq(rng, 300, max_len=90, k=9)
    for k, thresh in ((9, 38), (1, 50), (31, 38), (17, 0xC0)):
        for seq, qual in split_fastq(data):
            m, hq = po.extract_read(seq, qual, k, thresh)
            want = window_kmers(seq, qual, k, thresh)
            assert list(zip(m.tolist(), hq.tolist())) == want


def test_table_matches_dictionary_model():
    rng = np.random.default_rng(11)
    data = random_fastq(rng, 1500, k=15)
    for k, bits in ((15, 7), (15, 2), (24, 1)):
        inst = []
        for seq, qual in split_fastq(data):
            inst += window_kmers(seq, qual, k, 38)
        want = dict_model(inst, bits)
        db = po.OracleDB(256, k, bits)  # small: forces growth
        n = db.add_fastq(data, 38)
        assert n == len(inst)
        m, v = db.sorted_dump()
        assert dict(zip(m.tolist(), v.tolist())) == want
        um, uv = dict_model_np([a for a, _ in inst], [b for _, b in inst], bits)
        assert (um == m).all() and (uv == v).all()
        # multithreaded build gives the same content
        db2 = po.OracleDB(1 << 17, k, bits)
        assert db2.add_fastq(data, 38, threads=4) == n
        m2, v2 = db2.sorted_dump()
        assert (m2 == m).all() and (v2 == v).all()


def test_revcomp_canonical():
    rng = np.random.default_rng(5)
    for k in (1, 4, 24, 31):
        for m in rng.integers(0, 4 ** k, size=50).tolist():
            assert po.revcomp(m, k) == revcomp(m, k)
            assert po.canonical(m, k) == min(m, revcomp(m, k))


def test_parser_rules():
    """jellyfish::whole_sequence_parser FASTQ rules as the oracle restates them"""
    assert po.parse_fastq(b"") == []
    assert po.parse_fastq(b"@a\nACGT\n+\nIIII\n") == [(b"ACGT", b"IIII")]
    assert po.parse_fastq(b"@a\nACGT\n+\nIIII") == [(b"ACGT", b"IIII")]
    # multi-line sequence and quality; a quality line may start with '@' or '+'
    assert po.parse_fastq(b"@a\nAC\nGT\n+a\n@I\n+I\n@b\nAA\n+\nII\n") == [(b"ACGT", b"@I+I"), (b"AA", b"II")]
    for bad, msg in ((b"@a\nACGT\n", "Truncated"), (b"@a\nACGT\n+\nIII\n", "wrong number"),
                     (b"@a\nACGT\n+\nIIIII\n", "wrong number"), (b"@a\nACGT\n+\nIIII\nX\n", "header missing"),
                     (b">a\nACGT\n", "Unsupported"), (b"@a\nACGT\n+\nIIII\n\n", "header missing")):
        with pytest.raises(ValueError, match=msg):
            po.parse_fastq(bad)


def best_alt_model(table, fmer, backward, k):
    sh = 2 * (k - 1) if backward else 0
    counts, count, level, ucode = [0, 0, 0, 0], 0, 0, 0
    for i in range(4):
        m = (fmer & ~(3 << sh)) | (i << sh)
        v = table.get(min(m, revcomp(m, k)), 0)
        first, second = v >> 1, v & 1
        if first > 0 and second >= level:
            if second > level and count > 0:
                for j in range(i):
                    counts[j] = 0
                count = 0
            counts[i], ucode, level = first, i, second
            count += 1
    return count, counts, ucode, level


def test_lookup_semantics():
    rng = np.random.default_rng(13)
    k, bits = 7, 3  # 4^7 = 16384 possible mers: siblings collide often
    data = random_fastq(rng, 400, max_len=60, k=k)
    db = po.OracleDB(1 << 15, k, bits)
    db.add_fastq(data, 38)
    m, v = db.dump()
    table = dict(zip(m.tolist(), v.tolist()))
    for fm in rng.integers(0, 4 ** k, size=2000).tolist():
        c = min(fm, revcomp(fm, k))
        val = table.get(c, 0)
        assert db.lookup([c])[0] == val
        assert db.get_val(c) == ((val >> 1) if (val & 1) else 0)  # mer_database.hpp:296-299
        for bw in (0, 1):
            assert db.best_alternatives(fm, bw) == best_alt_model(table, fm, bw, k)
    d, t = db.stats()
    hqv = [x for x in table.values() if (x & 1) and x >= 2]
    assert (d, t) == (len(hqv), sum(x >> 1 for x in hqv))
    h = db.histo()
    assert int(h.sum()) == len(table)
    for x in set(table.values()):
        assert h[min(x >> 1, 1000), x & 1] == sum(1 for y in table.values() if y == x)


# ---- "next" rows: Poisson cutoff and the contaminant set --------------------------------
def test_poisson_term_known_answers():
    """error_correct_reads.cc:53-61 against values worked by hand: exact branch (i < 11) is
    e^-l l^i / i!; Stirling branch is within 1/(12 i) of it."""
    import math
    for lam, i in [(0.1, 0), (0.1, 2), (0.1, 3), (0.1, 4), (2.5, 7), (12.0, 10)]:
        want = math.exp(-lam) * lam ** i / math.factorial(i)
        assert po.poisson_term(lam, i) == pytest.approx(want, rel=1e-14)
    assert po.poisson_term(0.1, 2) == pytest.approx(0.0045241870901797976, rel=1e-13)
    for lam, i in [(12.0, 11), (12.0, 15), (3.0, 40)]:
        exact = math.exp(-lam) * lam ** i / math.factorial(i)
        st = po.poisson_term(lam, i)
        assert exact < st < exact * (1 + 1.0 / (11 * i))  # Stirling overestimates by ~1/(12 i)


def test_poisson_cutoff_known_answers():
    """error_correct_reads.cc:660-667 with the driver's defaults (quorum.in: apriori error rate
    0.01, poisson threshold 1e-6 => collision_prob 0.01/3, threshold 1e-4): coverage 30 gives
    lambda 0.1; terms 4.5e-3 (x=2), 1.5e-4 (x=3), 3.8e-6 (x=4) -> first below 1e-4 is x=4 -> 5."""
    assert po.poisson_cutoff_from(1000, 30000, 0.01 / 3, 1e-6 / 0.01) == 5
    # lambda = 1: terms e^-1/x! = 0.184, 0.0613, 0.0153, 3.07e-3, 5.1e-4, 7.3e-5 (x=7) -> 8
    assert po.poisson_cutoff_from(1, 300, 0.01 / 3, 1e-4) == 8
    # threshold never reached within x < 1000 -> 0 (:667)
    assert po.poisson_cutoff_from(1, 1, 0.5, 0.0) == 0
    db = po.OracleDB(1 << 10, 5, 7)
    for m, n in [(1, 30), (7, 30), (9, 30)]:
        for _ in range(n):
            db.add(m, 1)
    for _ in range(50):
        db.add(11, 0)  # low-quality mers do not enter the coverage estimate (:655)
    assert db.stats() == (3, 90)
    assert db.poisson_cutoff(0.01 / 3, 1e-4) == 5


def test_contaminant_set_from_adapter_fasta():
    """The reference's data/adapter.fa (copied as a data fixture to tests/golden/adapter.fa):
    the set must equal every canonical k-mer window of every record, computed here in pure
    Python. 'parity unpinned' against upstream's `jellyfish count -C` dump, which cannot be
    produced in this image."""
    path = os.path.join(os.path.dirname(__file__), "golden", "adapter.fa")
    data = open(path, "rb").read()
    k = 24
    want = set()
    inst = 0
    for rec in data.split(b">")[1:]:
        seq = b"".join(rec.split(b"\n")[1:])
        for m, _ in window_kmers(seq, b"I" * len(seq), k, 0):
            want.add(m)
            inst += 1
    db = po.OracleDB(1 << 15, k, 1)
    assert db.add_fasta(data) == inst
    mers, vals = db.sorted_dump()
    assert mers.tolist() == sorted(want)
    assert (vals & np.uint64(1)).all()
    # multi-line sequences and an N: same set as the joined line
    db2 = po.OracleDB(1 << 8, 4, 1)
    db2.add_fasta(b">a desc\nACGTA\nCGNAC\nGTT\n>b\nAC\n>c\nTTTTT")
    exp = {m for m, _ in window_kmers(b"ACGTACGNACGTT", b"I" * 13, 4, 0)} | {m for m, _ in window_kmers(b"TTTTT", b"IIIII", 4, 0)}
    assert set(db2.dump()[0].tolist()) == exp
    with pytest.raises(ValueError, match="Unsupported format"):
        db2.add_fasta(b"ACGT\n")
