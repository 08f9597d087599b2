"""GPU parity tests of the build path, through the C ABI (libquorum_hip.so), against
the CPU oracle and the hand-derived golden fixtures. Bit-exact: every
(k-mer, count, HQ bit) tuple must match, compared by content (sorted dump)."""
import ctypes as C
import os

import numpy as np
import pytest

import pyoracle as po
import quorum_amd as qa
from quorum_amd import _lib, synth
from helpers import golden_cases, mer_from_str, random_fastq, sorted_pairs, dict_model_np

pytestmark = pytest.mark.gpu


def gpu_build(data, k, bits, thresh, size=1 << 16, host_parsed=False, **kw):
    h = qa.HashWithQuality(size, 2 * k, bits, min_qual=thresh, **kw)
    if host_parsed:
        h.add_fastq_host_parsed(data, final=True)
    else:
        h.add_fastq(data)
    st = h.done()
    m, v = sorted_pairs(*h.dump())
    return h, st, m, v


def oracle_build(data, k, bits, thresh, size=1 << 16):
    db = po.OracleDB(size, k, bits)
    n = db.add_fastq(data, thresh)
    m, v = db.sorted_dump()
    return db, n, m, v


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_golden_known_answers(case):
    h, st, m, v = gpu_build(case["data"], case["k"], case["bits"], case["min_qual"], size=64)
    got = {int(a): (int(b) >> 1, int(b) & 1) for a, b in zip(m, v)}
    want = {mer_from_str(s): tuple(cv) for s, cv in case["expected"].items()}
    assert got == want
    assert st["distinct"] == len(want)
    # and the same through the sequential host parser entry point
    h2, st2, m2, v2 = gpu_build(case["data"], case["k"], case["bits"], case["min_qual"], size=64, host_parsed=True)
    assert (m2 == m).all() and (v2 == v).all()


@pytest.mark.parametrize("k,bits,thresh,seed", [(24, 7, 38, 1), (31, 7, 38, 2), (15, 2, 50, 3), (5, 1, 38, 4),
                                                (1, 3, 38, 5), (17, 7, 0xC0, 6), (31, 12, 38, 7)])
def test_random_fastq_matches_oracle(k, bits, thresh, seed):
    rng = np.random.default_rng(seed)
    data = random_fastq(rng, 3000, k=k)
    db, n, om, ov = oracle_build(data, k, bits, thresh, size=1 << 20)
    h, st, m, v = gpu_build(data, k, bits, thresh, size=1 << 20)
    assert st["instances"] == n and st["records"] == 3000 and st["distinct"] == om.size
    assert m.size == om.size and (m == om).all() and (v == ov).all()
    hq = int((np.asarray(v) & 1).sum())
    assert st["hq_instances"] >= hq


def test_growth_from_tiny_table_keeps_content():
    """-s far too small: the table doubles (handle_full_ary, mer_database.hpp:137-187); content unchanged"""
    rng = np.random.default_rng(21)
    data = random_fastq(rng, 4000, k=21)
    db, n, om, ov = oracle_build(data, 21, 7, 38, size=1 << 20)
    h, st, m, v = gpu_build(data, 21, 7, 38, size=100)
    assert st["grows"] >= 5 and st["slots"] >= om.size
    assert (m == om).all() and (v == ov).all()
    # several chunks, growing in between
    h2 = qa.HashWithQuality(16, 42, 7, min_qual=38)
    recs = data.split(b"@read")[1:]
    step = 500
    for i in range(0, len(recs), step):
        h2.add_fastq(b"".join(b"@read" + r for r in recs[i:i + step]))
    st2 = h2.done()
    m2, v2 = sorted_pairs(*h2.dump())
    assert st2["instances"] == n and (m2 == om).all() and (v2 == ov).all()


def test_reference_six_classes_write_read(tmp_path):
    """TEST_P(MerDatabase, WriteRead) (unit_tests/test_mer_database.cc:47-135) through the class mirror:
    concurrent-order-free inserts of hq2,hq1,lqhq,hqlq,lq1,lq2, write, read back with database_query."""
    rng = np.random.default_rng(7)
    k, bits, n = 31, 4, 100000
    names = ("hq2", "hq1", "lqhq", "hqlq", "lq1", "lq2")
    mers = {}
    for name in names:
        codes = rng.integers(0, 4, size=n)
        cnt = n - k + 1
        m = np.zeros(cnt, dtype=np.uint64)
        for i in range(k):
            m = (m << np.uint64(2)) | codes[i:i + cnt].astype(np.uint64)
        mers[name] = m  # NOT canonicalised, as upstream inserts them (:24-25)
    for param in (1, 10):
        database = qa.HashWithQuality(param * n, 2 * k, bits, 10)
        # one batch interleaving all ten "threads" so that the orders mix on the GPU
        plan = [("hq2", 1), ("hq2", 1), ("hq1", 1), ("lqhq", 0), ("lqhq", 1), ("hqlq", 1), ("hqlq", 0),
                ("lq1", 0), ("lq2", 0), ("lq2", 0)]
        allm = np.concatenate([mers[a] for a, _ in plan])
        allq = np.concatenate([np.full(mers[a].size, q, dtype=np.uint8) for a, q in plan])
        perm = rng.permutation(allm.size)
        assert database.add(allm[perm], allq[perm])
        st = database.done()
        if param == 1:
            assert st["grows"] >= 1
        path = str(tmp_path / ("mer_database_%d" % param))
        database.write(path)
        info = _lib.QhDbInfo()
        assert _lib.lib.qh_db_header_info(path.encode(), C.byref(info)) == 0
        assert info.offset % 8 == 0                                                  # :102
        assert os.path.getsize(path) == info.offset + info.key_bytes + info.value_bytes  # :103-106
        db = qa.DatabaseQuery(path)
        assert db.header()["bits"] == bits and db.header()["key_len"] == 2 * k       # :113-114
        expect = {"hq2": (2, 1), "hq1": (1, 1), "lqhq": (1, 1), "hqlq": (1, 1), "lq1": (1, 0), "lq2": (2, 0)}
        mer_map = {}
        for name, (c, q) in expect.items():                                           # :115-120
            v = db.lookup(mers[name])
            assert (v >> np.uint64(1) == c).all() and (v & np.uint64(1) == q).all(), name
            mer_map.update(zip(mers[name].tolist(), v.tolist()))
        dm, dv = db.dump()                                                             # :123-134
        assert dm.size == len(mer_map)
        got = dict(zip(dm.tolist(), dv.tolist()))
        assert got == mer_map
        assert db[int(mers["hq2"][0])] == (2, 1) and db.get_val(int(mers["lq2"][0])) == 0
        # vals image = atomic_bits_array layout of the same values
        raw = np.fromfile(path, dtype=np.uint64, offset=info.offset)
        slots = raw[:info.size]
        vimg = raw[info.size:]
        vb = bits + 1
        vpw = 64 // vb
        idx = np.arange(info.size, dtype=np.uint64)
        fields = (vimg[idx // np.uint64(vpw)] >> ((idx % np.uint64(vpw)) * np.uint64(vb))) & np.uint64((1 << vb) - 1)
        assert (fields == (slots & np.uint64((1 << vb) - 1))).all()
        db.close()
        database.close()


def test_format_fallback_and_errors():
    k, bits = 5, 7
    multi = b"@a\nACGTA\nCCGTA\n+a\nIIIII\nIIII#\n@b\nAAAAACC\n+\nIIIIIII\n"
    h = qa.HashWithQuality(64, 2 * k, bits, min_qual=38)
    h.add_fastq(multi)                       # fast path sees 10 lines / a bad layout
    with pytest.raises(qa.QhError) as ei:
        h.sync()
    assert ei.value.status == _lib.QH_EFORMAT
    assert h.stats()["instances"] == 0       # nothing of the chunk was inserted
    h.add_fastq_host_parsed(multi, final=True)
    st = h.done()
    db, n, om, ov = oracle_build(multi, k, bits, 38, size=64)
    m, v = sorted_pairs(*h.dump())
    assert st["instances"] == n == 6 + 3 and (m == om).all() and (v == ov).all()
    for bad, msg in ((b"@a\nACGT\n", "Truncated"), (b"@a\nACGT\n+\nIII\n", "wrong number of quals"),
                     (b"@a\nACGT\n+\nIIII\nX\n", "header missing"), (b">a\nACGT\n", "Unsupported format")):
        h2 = qa.HashWithQuality(64, 2 * k, bits, min_qual=38)
        with pytest.raises(qa.QhError) as ei:
            h2.add_fastq_host_parsed(bad, final=True)
        assert ei.value.status == _lib.QH_EFORMAT and msg in str(ei.value)
        with pytest.raises(ValueError):
            po.parse_fastq(bad)
    # fast-path rejections: quality shorter/longer than the sequence, missing '+', missing '@'
    for bad in (b"@a\nACGTA\n+\nIIII\n", b"@a\nACGTA\n+\nIIIIII\n", b"@a\nACGTA\n-\nIIIII\n", b"@a\nACGTA\n+\nIIIII\nb\nACGTA\n+\nIIIII\n",
                b"@a\nACGTA\n+\nIIIII\n\n"):
        h3 = qa.HashWithQuality(64, 2 * k, bits, min_qual=38)
        h3.add_fastq(bad)
        with pytest.raises(qa.QhError) as ei:
            h3.sync()
        assert ei.value.status == _lib.QH_EFORMAT
    # non-final chunks: a trailing partial record is left to the caller
    h4 = qa.HashWithQuality(64, 2 * k, bits, min_qual=38)
    part = b"@a\nACGTA\n+\nIIIII\n@b\nACG"
    assert h4.add_fastq_host_parsed(part, final=False) == 17
    assert h4.done()["instances"] == 1


def test_burst_of_tiny_records_takes_the_host_parser():
    """the GPU indexer keeps at most 4096 newlines per 16 KiB of text (lines of 4 bytes on average); a chunk whose records
    are long enough on average but holds a burst of 12-byte records is rejected by the fast path as a whole (nothing
    inserted) and goes through the sequential parser, like every other layout the fast path does not take"""
    k, bits = 3, 7
    rng = np.random.default_rng(11)
    long_recs = random_fastq(rng, 400, k=k, genome=rng.integers(0, 4, size=5000))
    burst = b"".join(b"@\n" + bytes([b"ACGT"[i & 3]]) * 3 + b"\n+\nIII\n" for i in range(3000))   # 12 bytes, 4 lines each
    data = long_recs + burst + long_recs
    assert len(data) / (data.count(b"\n") // 4) > 16                     # passes the chunk-wide record count check
    h = qa.HashWithQuality(1 << 12, 2 * k, bits, min_qual=38)
    h.add_fastq(data)
    with pytest.raises(qa.QhError) as ei:
        h.sync()
    assert ei.value.status == _lib.QH_EFORMAT and h.stats()["instances"] == 0
    h.add_fastq_host_parsed(data, final=True)
    st = h.done()
    db, n, om, ov = oracle_build(data, k, bits, 38, size=1 << 12)
    m, v = sorted_pairs(*h.dump())
    assert st["instances"] == n and (m == om).all() and (v == ov).all()


def test_empty_and_degenerate_inputs():
    h = qa.HashWithQuality(64, 48, 7, min_qual=38)
    h.add_fastq(b"")
    assert h.done()["instances"] == 0
    m, v = h.dump()
    assert m.size == 0
    assert h.add(np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint8))
    # one read far longer than a 64-base column block, all one k-mer (worst-case CAS contention)
    seq = b"A" * 5000
    data = b"@poly\n" + seq + b"\n+\n" + b"I" * 5000 + b"\n"
    h, st, m, v = gpu_build(data, 24, 7, 38)
    assert st["instances"] == 5000 - 23 and m.tolist() == [0] and v.tolist() == [(127 << 1) | 1]
    h, st, m, v = gpu_build(data, 24, 12, 38)
    assert v.tolist() == [(min(4977, 4095) << 1) | 1]


def test_add_mers_matches_oracle_rule():
    rng = np.random.default_rng(31)
    k, bits = 12, 3
    mers = rng.integers(0, 4 ** k, size=200000, dtype=np.uint64)
    mers[::3] = mers[1::3][:mers[::3].size]  # duplicates
    hq = (rng.random(mers.size) < 0.4).astype(np.uint8)
    h = qa.HashWithQuality(1 << 12, 2 * k, bits)
    assert h.add(mers, hq)
    um, uv = dict_model_np(mers, hq, bits)
    m, v = sorted_pairs(*h.dump())
    assert (m == um).all() and (v == uv).all()
    db = po.OracleDB(1 << 12, k, bits)
    db.add_mers(mers, hq)
    om, ov = db.sorted_dump()
    assert (m == om).all() and (v == ov).all()


def test_full_size_config_properties():
    """BASELINE config 2 (5 Mbp genome, 2M x 100 bp, k=24, -s 50M, -b 7, -q 38) at full size:
    size-independent properties + oracle parity on the first 100k reads."""
    k, bits, thresh, L, n_reads = 24, 7, 38, 100, 2_000_000
    g = synth.make_genome(5_000_000)
    fq = synth.make_fastq(g, n_reads, L)
    rb = synth.record_bytes(L)
    data = fq.tobytes()
    h = qa.HashWithQuality(50_000_000, 2 * k, bits, min_qual=thresh)
    h.add_fastq(data)
    st = h.done()
    assert st["slots"] == 1 << 26 and st["grows"] == 0 and st["records"] == n_reads
    # every window of 24 valid bases is one instance: count them independently on the host
    bases = fq.reshape(n_reads, rb)[:, 12:12 + L]
    valid = np.isin(bases, np.frombuffer(b"ACGT", dtype=np.uint8))
    run = np.zeros(n_reads, dtype=np.int64)
    inst = 0
    for j in range(L):
        run = np.where(valid[:, j], run + 1, 0)
        inst += int((run >= k).sum())
    assert st["instances"] == inst
    m, v = h.dump()
    assert m.size == st["distinct"] == np.unique(m).size
    total = int((v >> np.uint64(1)).sum())
    assert total <= inst
    # order independence: reversed read order, in 3 chunks -> identical content
    recs = fq.reshape(n_reads, rb)[::-1]
    h2 = qa.HashWithQuality(50_000_000, 2 * k, bits, min_qual=thresh)
    for part in np.array_split(recs, 3):
        h2.add_fastq(np.ascontiguousarray(part).tobytes())
    st2 = h2.done()
    m2, v2 = sorted_pairs(*h2.dump())
    m, v = sorted_pairs(m, v)
    assert st2["instances"] == inst and (m2 == m).all() and (v2 == v).all()
    # inserting everything a second time: counts double (saturating), HQ bits unchanged
    h2.add_fastq(data)
    h2.done()
    m3, v3 = sorted_pairs(*h2.dump())
    c, q = v >> np.uint64(1), v & np.uint64(1)
    assert (m3 == m).all() and ((v3 & np.uint64(1)) == q).all()
    assert ((v3 >> np.uint64(1)) == np.minimum(2 * c, np.uint64(127))).all()
    h2.close()
    # oracle parity on a prefix the CPU finishes in seconds
    sub = fq[:100_000 * rb].tobytes()
    db = po.OracleDB(1 << 24, k, bits)
    n = db.add_fastq(sub, thresh, threads=8)
    om, ov = db.sorted_dump()
    h.reset()
    h.add_fastq(sub)
    assert h.done()["instances"] == n
    sm, sv = sorted_pairs(*h.dump())
    assert (sm == om).all() and (sv == ov).all()
