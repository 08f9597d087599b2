"""Early preparation (qh_build_add_fastq_device on the handle's own stream): while chunk i is still being split, the
index and segment kernels of chunk i+1 run on a second stream into the other set of per-chunk arrays. Content and
counters must be what the same chunks give one at a time (QH_NO_EARLY_PREP) and what the oracle gives; chunk sizes rise
and fall so that both sets of arrays are grown while the other one is in use."""
import numpy as np
import pytest
import torch

import pyoracle as po
import quorum_amd as qa
from quorum_amd import _lib, synth
from helpers import sorted_pairs

pytestmark = pytest.mark.gpu

K, BITS, Q, L = 24, 7, 38, 100
SLOTS = 1 << 22
RB = synth.record_bytes(L)
CHUNK_READS = (30_000, 50_000, 10_000, 50_000, 70_000, 5_000, 40_000)


@pytest.fixture(scope="module")
def reads():
    g = synth.make_genome(400_000, seed=41)
    return synth.make_fastq(g, sum(CHUNK_READS), L, err=0.001, seed=42, tails=True)


def device_chunks(fq, counts):
    out, at = [], 0
    for n in counts:
        out.append(torch.from_numpy(fq[at * RB:(at + n) * RB].copy()).to("cuda:0"))
        at += n
    torch.cuda.synchronize()      # the handle's own streams do not see torch's
    return out


def run(chunks, slots, defer, between=None):
    h = qa.HashWithQuality(slots, 2 * K, BITS, min_qual=Q)
    h.set_mode("partition")
    h.set_defer(defer)
    h.enable_timing(True)
    for i, c in enumerate(chunks):
        h.add_fastq_device(c.data_ptr(), c.numel())
        if between:
            between(h, i)
    st = h.done()
    st["timed_preps"] = h.kernel_time(0)[1]
    m, v = sorted_pairs(*h.dump())
    h.close()
    return st, m, v, st.pop("early_preps")


@pytest.mark.parametrize("defer", [0, 1 << 30])
def test_back_to_back_chunks_prepare_early(reads, defer, monkeypatch):
    chunks = device_chunks(reads, CHUNK_READS)
    db = po.OracleDB(2 * SLOTS, K, BITS)
    n = db.add_fastq(reads.tobytes(), Q, threads=8)
    om, ov = db.sorted_dump()
    st, m, v, timed = run(chunks, SLOTS, defer)
    assert timed == len(chunks) - 1        # all chunks but the first were prepared beside their predecessor
    assert st["timed_preps"] == len(chunks)   # early or not, the segment kernels of every chunk are timed
    assert st["instances"] == n and st["records"] == sum(CHUNK_READS) and st["distinct"] == om.size
    assert (m == om).all() and (v == ov).all()
    monkeypatch.setenv("QH_NO_EARLY_PREP", "1")
    st0, m0, v0, timed0 = run(chunks, SLOTS, defer)
    assert timed0 == 0
    assert st0 == st and (m0 == m).all() and (v0 == v).all()


def test_early_prep_through_growth_and_other_entry_points(reads):
    """a table that doubles under the chunks (the collection of chunk i, which runs after chunk i+1 was prepared,
    changes the geometry), item lists between the chunks (the chunk in flight is then not a FASTQ one) and a
    collection in the middle"""
    chunks = device_chunks(reads, CHUNK_READS)
    rng = np.random.default_rng(9)
    big = rng.integers(0, 1 << 48, size=300_000, dtype=np.uint64)
    big[::2] = big[1::2]
    big_q = (rng.random(big.size) < 0.3).astype(np.uint8)
    items = torch.from_numpy(((big << np.uint64(1)) | big_q).view(np.int64)).cuda()
    torch.cuda.synchronize()

    def between(h, i):
        if i == 1:
            h.add_mers_device(items.data_ptr(), big.size)
        if i == 3:
            h.sync()

    db = po.OracleDB(2 * SLOTS, K, BITS)
    n = db.add_fastq(reads.tobytes(), Q, threads=8)
    db.add_mers(big, big_q)
    om, ov = db.sorted_dump()
    st, m, v, timed = run(chunks, 1 << 19, 1 << 30, between)
    assert st["grows"] >= 2 and 1 <= timed < len(chunks)   # (not after the collection in the middle)
    assert st["instances"] == n + big.size and (m == om).all() and (v == ov).all()


def test_malformed_chunk_prepared_early_is_rejected_whole(reads):
    """the format verdict of an early-prepared chunk reaches the split kernels and the host as it does otherwise:
    nothing of the chunk is inserted, the error names the chunk at its collection, the chunks before it stay"""
    good = device_chunks(reads, CHUNK_READS[:2])
    bad_np = reads[:20_000 * RB].copy()
    bad_np[5 * RB + 3] = ord("\n")          # an extra newline inside a header: the line count is no longer 4 per record
    bad = torch.from_numpy(bad_np).to("cuda:0")
    torch.cuda.synchronize()
    h = qa.HashWithQuality(SLOTS, 2 * K, BITS, min_qual=Q)
    h.set_mode("partition")
    h.add_fastq_device(good[0].data_ptr(), good[0].numel())
    h.add_fastq_device(bad.data_ptr(), bad.numel())       # prepared while good[0] is split
    with pytest.raises(qa.QhError) as ei:
        h.sync()
    assert ei.value.status == _lib.QH_EFORMAT
    db = po.OracleDB(2 * SLOTS, K, BITS)
    n = db.add_fastq(reads[:CHUNK_READS[0] * RB].tobytes(), Q, threads=8)
    om, ov = db.sorted_dump()
    m, v = sorted_pairs(*h.dump())
    assert h.stats()["instances"] == n and (m == om).all() and (v == ov).all()
    h.close()


def test_builds_back_to_back_in_one_handle(reads):
    """flush() starts the pass over the table and returns, reset() with that pass in flight only notes itself, and the
    next build's first chunk is prepared beside the pass: three builds in a row, each collected by the one after it, the
    last by done(). Every build's table is the oracle's for its own reads alone; a reset noted and then followed by a
    read of the statistics shows an empty handle."""
    chunks = device_chunks(reads, CHUNK_READS)
    groups = [(0, 3), (3, 5), (5, 7)]
    h = qa.HashWithQuality(SLOTS, 2 * K, BITS, min_qual=Q)
    h.set_mode("partition")
    h.set_defer(1 << 30)
    first = 0
    for gi, (a, b) in enumerate(groups):
        if gi:
            h.reset()                               # the pass of the build before is still running
        for c in chunks[a:b]:
            h.add_fastq_device(c.data_ptr(), c.numel())
        if gi < len(groups) - 1:
            h.flush()
    st = h.done()
    a, b = groups[-1]
    lo, hi = sum(CHUNK_READS[:a]) * RB, sum(CHUNK_READS[:b]) * RB
    db = po.OracleDB(2 * SLOTS, K, BITS)
    n = db.add_fastq(reads[lo:hi].tobytes(), Q, threads=8)
    om, ov = db.sorted_dump()
    m, v = sorted_pairs(*h.dump())
    assert st["instances"] == n and st["records"] == sum(CHUNK_READS[a:b]) and st["distinct"] == om.size
    assert (m == om).all() and (v == ov).all()
    assert st["early_preps"] == b - a               # the build's first chunk too: prepared beside the pass before it
    # the middle build, looked at before the reset that follows it is applied
    h.reset()
    a, b = groups[1]
    for c in chunks[a:b]:
        h.add_fastq_device(c.data_ptr(), c.numel())
    h.flush()
    lo, hi = sum(CHUNK_READS[:a]) * RB, sum(CHUNK_READS[:b]) * RB
    db = po.OracleDB(2 * SLOTS, K, BITS)
    n = db.add_fastq(reads[lo:hi].tobytes(), Q, threads=8)
    om, ov = db.sorted_dump()
    m, v = sorted_pairs(*h.dump())
    assert h.stats()["instances"] == n and (m == om).all() and (v == ov).all()
    h.add_fastq_device(chunks[0].data_ptr(), chunks[0].numel())
    h.reset()                                       # noted: the chunk is in flight
    st = h.stats()                                  # collects it, then empties the handle
    assert st["instances"] == 0 and st["distinct"] == 0 and st["records"] == 0
    m, v = h.dump()
    assert m.size == 0
    h.close()


def test_reset_swallows_the_format_error_of_the_chunk_in_flight(reads):
    bad_np = reads[:20_000 * RB].copy()
    bad_np[5 * RB + 3] = ord("\n")
    bad = torch.from_numpy(bad_np).to("cuda:0")
    good = device_chunks(reads, CHUNK_READS[:1])[0]
    torch.cuda.synchronize()
    h = qa.HashWithQuality(SLOTS, 2 * K, BITS, min_qual=Q)
    h.set_mode("partition")
    h.add_fastq_device(bad.data_ptr(), bad.numel())
    h.reset()                                       # eager or noted: the rejected chunk goes with the rest
    h.add_fastq_device(good.data_ptr(), good.numel())
    st = h.done()
    db = po.OracleDB(2 * SLOTS, K, BITS)
    n = db.add_fastq(reads[:CHUNK_READS[0] * RB].tobytes(), Q, threads=8)
    assert st["instances"] == n and st["records"] == CHUNK_READS[0]
    h.close()
