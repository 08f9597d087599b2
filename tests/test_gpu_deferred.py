"""GPU parity tests of the deferred window pass (qh_build_set_defer): the window buckets of successive partitioned
chunks accumulate in HBM and the table is streamed once per batch. Whatever the budget, whatever is interleaved with
the chunks (lookups on the live table, dumps, resets, direct-path inserts, growth, spills), the table's content is the
oracle's -- hash_with_quality::add is order independent (src/mer_database.hpp:102-111) -- bit for bit."""
import numpy as np
import pytest
import torch

import pyoracle as po
import quorum_amd as qa
from quorum_amd import _lib, synth
from helpers import sorted_pairs, dict_model_np

pytestmark = pytest.mark.gpu

K, BITS, Q, L = 24, 7, 38, 100
SLOTS = 1 << 22          # 512 windows: the smallest table class that takes the two-level partitioned build
RB = synth.record_bytes(L)


@pytest.fixture(scope="module")
def reads():
    g = synth.make_genome(400_000, seed=11)
    fq = synth.make_fastq(g, 600_000, L, err=0.001, seed=12, tails=True)   # ~1.6 M distinct mers: load 0.4
    return fq


def oracle_of(fq_bytes, size=2 * SLOTS):
    db = po.OracleDB(size, K, BITS)
    n = db.add_fastq(fq_bytes, Q, threads=8)
    return db, n


def feed(h, fq, n_chunks, first=0, per=30_000):
    for c in range(first, first + n_chunks):
        h.add_fastq(fq[c * per * RB:(c + 1) * per * RB].tobytes())


def test_batches_stream_the_table_once_and_match_the_oracle(reads):
    db, n = oracle_of(reads.tobytes())
    om, ov = db.sorted_dump()
    results = {}
    for budget in (0, 1 << 30, -1):
        h = qa.HashWithQuality(SLOTS, 2 * K, BITS, min_qual=Q)
        h.set_mode("partition")
        h.set_defer(budget)
        feed(h, reads, 20)
        st = h.done()
        m, v = sorted_pairs(*h.dump())
        assert st["instances"] == n and st["records"] == 600_000 and st["distinct"] == om.size
        assert (m == om).all() and (v == ov).all(), budget
        results[budget] = st["window_passes"]
        h.close()
    assert results[0] == 20                 # every chunk pays for a pass over the table
    assert 1 <= results[1 << 30] <= 5       # 2.3 M items a chunk; 15-bit counters beside the 34-bit tag: ~14 M a batch
    assert results[-1] <= 20


def test_interleaved_queries_dumps_and_resets(reads):
    per = 30_000
    h = qa.HashWithQuality(SLOTS, 2 * K, BITS, min_qual=Q)
    h.set_mode("partition")
    h.set_defer(1 << 30)
    feed(h, reads, 3)
    # a lookup on the live table must see the three chunks although no flush was asked for
    db3, n3 = oracle_of(reads[:3 * per * RB].tobytes())
    om, ov = db3.sorted_dump()
    q = h.as_query()
    assert (q.lookup(om) == ov).all()
    assert q.stats() == db3.stats()
    q.close()
    assert h.stats()["window_passes"] == 1
    feed(h, reads, 2, first=3)
    m, v = sorted_pairs(*h.dump())           # dump flushes too
    db5, n5 = oracle_of(reads[:5 * per * RB].tobytes())
    om5, ov5 = db5.sorted_dump()
    assert (m == om5).all() and (v == ov5).all() and h.stats()["instances"] == n5
    # reset with chunks waiting in the buckets: they are dropped with the table
    feed(h, reads, 2, first=5)
    h.reset()
    feed(h, reads, 4, first=10)
    st = h.done()
    dbr, nr = oracle_of(reads[10 * per * RB:14 * per * RB].tobytes())
    omr, ovr = dbr.sorted_dump()
    m, v = sorted_pairs(*h.dump())
    assert st["instances"] == nr and st["window_passes"] == 1
    assert (m == omr).all() and (v == ovr).all()
    # the no-allocation entry point refuses to run beside an open batch
    feed(h, reads, 1, first=14)
    items = torch.from_numpy(((om[:1000] << np.uint64(1)) | np.uint64(1)).view(np.int64)).cuda()
    with pytest.raises(qa.QhError):
        h.add_mers_device_noalloc(items.data_ptr(), 1000)
    h.sync()
    h.add_mers_device_noalloc(items.data_ptr(), 1000)
    h.close()


def test_direct_inserts_between_deferred_chunks(reads):
    """mers handed over as items (small: the direct global-CAS path) between chunks that wait in the buckets; large item
    lists join the batch"""
    per = 30_000
    rng = np.random.default_rng(3)
    h = qa.HashWithQuality(SLOTS, 2 * K, BITS, min_qual=Q)
    h.set_defer(1 << 30)                      # automatic strategy: chunks of 2.3 M items open a batch (slots / 64 = 65 K)
    feed(h, reads, 2)
    extra = rng.integers(0, 1 << 48, size=5000, dtype=np.uint64)
    extra_q = (rng.random(5000) < 0.5).astype(np.uint8)
    assert h.add(extra, extra_q)              # flushes the batch first, then inserts one by one
    feed(h, reads, 2, first=2)
    big = rng.integers(0, 1 << 48, size=400_000, dtype=np.uint64)
    big[::2] = big[1::2]
    big_q = (rng.random(big.size) < 0.3).astype(np.uint8)
    items = torch.from_numpy(((big << np.uint64(1)) | big_q).view(np.int64)).cuda()
    torch.cuda.synchronize()
    h.add_mers_device(items.data_ptr(), big.size)
    feed(h, reads, 1, first=4)
    st = h.done()
    db, n = oracle_of(reads[:5 * per * RB].tobytes())
    db.add_mers(extra, extra_q)
    db.add_mers(big, big_q)
    om, ov = db.sorted_dump()
    m, v = sorted_pairs(*h.dump())
    assert st["instances"] == n + 5000 + big.size
    assert (m == om).all() and (v == ov).all()
    assert st["window_passes"] <= 3
    h.close()


def test_a_bucket_that_overflows_spills_and_flushes():
    """100 000 copies of one read put 7.7 M items into one window's bucket (65 K at most): the rest takes the spill
    list, which forces the batch out at the next collection; counts saturate at 127"""
    g = synth.make_genome(200_000, seed=5)
    fq = synth.make_fastq(g, 60_000, L, err=0.0, n_rate=0.0, seed=6)
    one = fq[:RB].tobytes()
    h = qa.HashWithQuality(SLOTS, 2 * K, BITS, min_qual=Q)
    h.set_mode("partition")
    h.set_defer(1 << 30)
    h.add_fastq(fq.tobytes())
    h.add_fastq(one * 100_000)
    h.add_fastq(fq.tobytes())
    st = h.done()
    db, n = oracle_of(fq.tobytes() + one * 1000 + fq.tobytes())   # 1000 copies saturate the oracle's counts as well
    om, ov = db.sorted_dump()
    m, v = sorted_pairs(*h.dump())
    assert st["instances"] == n + 99_000 * (L - K + 1)
    assert (m == om).all() and (v == ov).all()
    assert st["window_passes"] >= 2
    h.close()


def test_growth_inside_and_between_batches():
    """a table far too small for its reads: the flush overflows, the table doubles (handle_full_ary,
    src/mer_database.hpp:137-187), later chunks open a batch for the new geometry"""
    g = synth.make_genome(3_000_000, seed=21)
    fq = synth.make_fastq(g, 300_000, L, err=0.01, seed=22)
    h = qa.HashWithQuality(1 << 21, 2 * K, BITS, min_qual=Q)
    h.set_mode("partition")
    h.set_defer(1 << 30)
    per = 50_000
    for c in range(6):
        h.add_fastq(fq[c * per * RB:(c + 1) * per * RB].tobytes())
        if c == 2:
            h.sync()
    st = h.done()
    db, n = oracle_of(fq.tobytes(), size=1 << 24)
    om, ov = db.sorted_dump()
    m, v = sorted_pairs(*h.dump())
    assert st["grows"] >= 2 and st["slots"] >= om.size
    assert st["instances"] == n and (m == om).all() and (v == ov).all()
    h.close()


def test_sharded_handles_defer_alike(reads):
    per = 30_000
    db, n = oracle_of(reads[:8 * per * RB].tobytes())
    om, ov = db.sorted_dump()
    parts = []
    for s in (0, 1):
        h = qa.HashWithQuality(2 * SLOTS, 2 * K, BITS, min_qual=Q, shard_bits=1, shard_id=s)
        h.set_mode("partition")
        h.set_defer(1 << 30)
        feed(h, reads, 8)
        st = h.done()
        assert st["window_passes"] == 1 and st["instances"] + st["foreign"] == n
        parts.append(h.dump())
        h.close()
    m, v = sorted_pairs(np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]))
    assert (m == om).all() and (v == ov).all()
