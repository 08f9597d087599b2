"""The CPU oracle against the reference's own known answers
(unit_tests/test_mer_database.cc:115-120), the hand-derived fixtures of
tests/golden/  and two independent models. CPU only."""
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
