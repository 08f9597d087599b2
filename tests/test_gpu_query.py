"""GPU parity tests of the database_query side (batched lookups, get_val,
get_best_alternatives, table reductions, file round trip) against the CPU oracle."""
import numpy as np
import pytest

import pyoracle as po
import quorum_amd as qa
from helpers import random_fastq, sorted_pairs, revcomp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small_db(tmp_path_factory):
    rng = np.random.default_rng(17)
    k, bits = 9, 5  # 4^9 = 262144 possible mers: sibling lookups hit often
    data = random_fastq(rng, 6000, max_len=120, k=k, genome=rng.integers(0, 4, size=40000))
    db = po.OracleDB(1 << 18, k, bits)
    db.add_fastq(data, 38)
    h = qa.HashWithQuality(1 << 18, 2 * k, bits, min_qual=38)
    h.add_fastq(data)
    h.done()
    path = str(tmp_path_factory.mktemp("db") / "q.db")
    h.write(path, argv=["test", "-s", "256k"])
    return dict(k=k, bits=bits, oracle=db, build=h, path=path, rng=rng)


@pytest.mark.parametrize("source", ["file", "live"])
def test_lookup_get_val(small_db, source):
    k, db = small_db["k"], small_db["oracle"]
    q = qa.DatabaseQuery(small_db["path"]) if source == "file" else small_db["build"].as_query()
    assert (q.k, q.bits) == (k, small_db["bits"])
    rng = np.random.default_rng(3)
    mers = rng.integers(0, 4 ** k, size=50000, dtype=np.uint64)
    canon = np.array([min(m, revcomp(m, k)) for m in mers.tolist()], dtype=np.uint64)
    want = db.lookup(canon)
    got = q.lookup(canon)
    assert (got == want).all()
    assert (want != 0).sum() > 1000 and (want == 0).sum() > 1000  # both hits and misses are exercised
    gv = q.get_val(canon)
    assert (gv == np.where(want & np.uint64(1), want >> np.uint64(1), np.uint64(0))).all()
    om, ov = db.sorted_dump()
    m, v = sorted_pairs(*q.dump())
    assert (m == om).all() and (v == ov).all()
    assert (q.lookup(om) == ov).all()
    # iteration protocol of database_query::begin()/end()
    it = dict(iter(q))
    assert len(it) == om.size and it[int(om[0])] == (int(ov[0]) >> 1, int(ov[0]) & 1)
    q.close()


def test_best_alternatives(small_db):
    k, db = small_db["k"], small_db["oracle"]
    q = qa.DatabaseQuery(small_db["path"])
    rng = np.random.default_rng(5)
    fm = rng.integers(0, 4 ** k, size=20000, dtype=np.uint64)
    bw = (rng.random(fm.size) < 0.5).astype(np.uint8)
    nkept, counts, ucode, level = q.get_best_alternatives(fm, bw)
    kept_hist = np.zeros(5, dtype=int)
    for i in range(fm.size):
        n, c, u, l = db.best_alternatives(int(fm[i]), int(bw[i]))
        assert (int(nkept[i]), counts[i].tolist(), int(ucode[i]), int(level[i])) == (n, c, u, l), i
        kept_hist[n] += 1
    assert kept_hist[0] > 0 and kept_hist[1] > 0 and kept_hist[2:].sum() > 0
    q.close()


def test_stats_and_histogram(small_db):
    db = small_db["oracle"]
    q = qa.DatabaseQuery(small_db["path"])
    assert q.stats() == db.stats()
    assert (q.histo() == db.histo()).all()
    q.close()


def test_open_errors(tmp_path):
    with pytest.raises(RuntimeError, match="Can't open"):
        qa.DatabaseQuery(str(tmp_path / "nope"))
    p = tmp_path / "junk"
    p.write_bytes(b"junk" * 10)
    with pytest.raises(RuntimeError, match="Can't parse header"):
        qa.DatabaseQuery(str(p))
