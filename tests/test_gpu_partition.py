"""The radix-partitioned build (qh_partition.hip) against the oracle and against the
direct global-CAS build: both strategies must give the same content on every input."""
import numpy as np
import pytest
import torch

import pyoracle as po
import quorum_amd as qa
from quorum_amd import synth
from helpers import random_fastq, sorted_pairs, dict_model_np

pytestmark = pytest.mark.gpu

# Liveness guard of the window kernels. A claim that is withdrawn and retried in the NEXT round instead of in place
# lets two waves in step fail each other's compare-and-swap for seconds (launches of 5 s and 23 s were measured,
# DESIGN section 6) while every result stays correct: a parity test alone would pass, slowly. At these sizes a
# healthy window pass takes well under a millisecond per chunk.
WINDOW_PASS_BOUND_MS = 50.0


def assert_window_pass_is_live(h, what):
    ms, launches = h.kernel_time(3)
    assert launches > 0 and ms < WINDOW_PASS_BOUND_MS, "%s: %d window passes took %.1f ms" % (what, launches, ms)


def build(data, k, bits, size, mode, level1_bits=-1, chunks=1, thresh=38, **kw):
    h = qa.HashWithQuality(size, 2 * k, bits, min_qual=thresh, **kw)
    h.set_mode(mode, level1_bits)
    h.enable_timing(True)
    if chunks == 1:
        h.add_fastq(data)
    else:
        recs = data.split(b"@read")[1:]
        step = (len(recs) + chunks - 1) // chunks
        for i in range(0, len(recs), step):
            h.add_fastq(b"".join(b"@read" + r for r in recs[i:i + step]))
    st = h.done()
    m, v = sorted_pairs(*h.dump())
    return h, st, m, v


# (1 << 24, 10, 2): 1024 level-1 bins in front of a second level, as the 2^32-slot tables have
@pytest.mark.parametrize("size,level1_bits,chunks", [(1 << 14, -1, 1), (1 << 16, -1, 1), (1 << 20, -1, 3),
                                                     (1 << 20, 3, 1), (1 << 22, 4, 2), (1 << 22, 0, 1),
                                                     (1 << 24, 10, 2)])
@pytest.mark.parametrize("k,bits", [(24, 7), (31, 3), (9, 5)])
def test_partitioned_matches_oracle(size, level1_bits, chunks, k, bits):
    rng = np.random.default_rng(size % 1000 + k)
    data = random_fastq(rng, 5000, k=k, genome=rng.integers(0, 4, size=30000))
    db = po.OracleDB(1 << 22, k, bits)
    n = db.add_fastq(data, 38, threads=4)
    om, ov = db.sorted_dump()
    h, st, m, v = build(data, k, bits, size, "partition", level1_bits, chunks)
    assert st["instances"] == n and st["records"] == 5000 and st["distinct"] == om.size
    assert m.size == om.size and (m == om).all() and (v == ov).all()
    h2, st2, m2, v2 = build(data, k, bits, size, "direct")
    assert (m2 == om).all() and (v2 == ov).all()


@pytest.mark.parametrize("k,bits,size", [(12, 2, 1 << 18), (16, 7, 1 << 20), (24, 3, 1 << 26), (31, 3, 1 << 26)])
def test_window_kernels_agree(k, bits, size, monkeypatch):
    """the counting form of the window build ([tag | H | L] words, one add per sighting) against the
    compare-and-swap form and the oracle: deep coverage (saturating counts), mixed qualities, three
    chunks so that windows are converted from and to the table's form with content in them"""
    rng = np.random.default_rng(k * 100 + bits)
    data = random_fastq(rng, 9000, k=k, genome=rng.integers(0, 4, size=3000))
    db = po.OracleDB(1 << 20, k, bits)
    n = db.add_fastq(data, 38, threads=4)
    om, ov = db.sorted_dump()
    assert (ov >> np.uint64(1)).max() == (1 << bits) - 1  # saturation is reached
    assert 0 < (ov & np.uint64(1)).sum() < ov.size        # both quality levels present
    # the default (16-bit counters where tag and bucket allow; k = 31: wide tags, 10-bit counters), generic counters, CAS
    for kernel in ("count16", "count", "cas"):
        monkeypatch.setenv("QH_WINDOW_KERNEL", kernel)
        h, st, m, v = build(data, k, bits, size, "partition", chunks=3)
        assert st["instances"] == n and st["distinct"] == om.size
        assert (m == om).all() and (v == ov).all(), kernel
        assert_window_pass_is_live(h, "QH_WINDOW_KERNEL=" + kernel)
        h.close()


def test_watched_counters_for_wide_tags(monkeypatch):
    """k = 31 into 2^28 slots: 42 tag bits leave 11-bit counters, fewer than a bucket of this chunk holds (3.9 K). The
    counting kernel then watches its counters (GUARD): a window where a key is seen a thousand times -- here the one
    with 360 000 poly-A sightings -- is left as it was and run again by the compare-and-swap kernel; every other
    window counts. Same content as the compare-and-swap kernel alone and as the oracle."""
    from quorum_amd import synth
    k, bits, L = 31, 7, 150
    g = synth.make_genome(2_000_000, seed=31)
    fq = synth.make_fastq(g, 600_000, L, err=0.01, seed=32, tails=True).tobytes()
    fq += b"".join(b"@polyA%d\n" % i + b"A" * L + b"\n+\n" + b"I" * L + b"\n" for i in range(3000))
    db = po.OracleDB(1 << 27, k, bits)
    n = db.add_fastq(fq, 38, threads=8)
    om, ov = db.sorted_dump()
    passes = {}
    for kernel in ("", "noguard"):
        monkeypatch.setenv("QH_WINDOW_KERNEL", kernel)
        h = qa.HashWithQuality(1 << 28, 2 * k, bits, min_qual=38)
        h.set_mode("partition")
        h.enable_timing(True)
        h.add_fastq(fq)
        st = h.done()
        m, v = sorted_pairs(*h.dump())
        assert st["instances"] == n and st["distinct"] == om.size
        assert (m == om).all() and (v == ov).all(), kernel
        passes[kernel] = st["window_passes"]
        assert_window_pass_is_live(h, "watched counters / " + kernel)
        h.close()
    assert passes == {"": 2, "noguard": 1}   # the hot window went round again, at the same table size


@pytest.mark.parametrize("k,bits,size,chunks", [(12, 7, 1 << 14, 3), (16, 3, 1 << 20, 2), (9, 5, 1 << 16, 1), (17, 12, 1 << 22, 2)])
def test_packed_window_items(k, bits, size, chunks, monkeypatch):
    """short remainders (2k + 1 - log2(slots) <= 18): the window buckets hold 32-bit items {remainder | position |
    quality}. Same table as with 64-bit items (QH_NO_PACK) and as the oracle's, through every window kernel, deferred
    batches, and -- the (12, 7, 2^14) case -- a table that doubles four times under its reads, whose failed windows are
    run again from buckets packed for the smaller table."""
    rng = np.random.default_rng(900 + k)
    data = random_fastq(rng, 12000, k=k, genome=rng.integers(0, 4, size=150000))
    db = po.OracleDB(1 << 22, k, bits)
    n = db.add_fastq(data, 38, threads=4)
    om, ov = db.sorted_dump()
    for no_pack in (False, True):
        if no_pack:
            monkeypatch.setenv("QH_NO_PACK", "1")
        for kernel in ("", "count", "cas"):
            monkeypatch.setenv("QH_WINDOW_KERNEL", kernel)
            for defer in (0, 1 << 28):
                h = qa.HashWithQuality(size, 2 * k, bits, min_qual=38)
                h.set_mode("partition")
                h.set_defer(defer)
                recs = data.split(b"@read")[1:]
                step = (len(recs) + chunks - 1) // chunks
                for i in range(0, len(recs), step):
                    h.add_fastq(b"".join(b"@read" + r for r in recs[i:i + step]))
                st = h.done()
                m, v = sorted_pairs(*h.dump())
                assert st["instances"] == n and st["distinct"] == om.size, (no_pack, kernel, defer)
                assert (m == om).all() and (v == ov).all(), (no_pack, kernel, defer)
                if size == 1 << 14:
                    assert st["grows"] >= 3
                h.close()


def test_partitioned_growth_and_contention():
    """a 16384-slot table receiving far more distinct k-mers than it holds: windows overflow, the
    spill list is inserted directly, the table doubles; plus a poly-A read (one slot, 5000 updates)"""
    rng = np.random.default_rng(77)
    data = random_fastq(rng, 6000, k=21, genome=rng.integers(0, 4, size=200000))
    data += b"@readpoly\n" + b"A" * 5000 + b"\n+\n" + b"I" * 5000 + b"\n"
    db = po.OracleDB(1 << 22, 21, 7)
    n = db.add_fastq(data, 38, threads=4)
    om, ov = db.sorted_dump()
    assert om.size > 4 * (1 << 14)
    h, st, m, v = build(data, 21, 7, 1 << 14, "partition", chunks=2)
    assert st["instances"] == n and st["grows"] >= 2
    assert (m == om).all() and (v == ov).all()
    assert_window_pass_is_live(h, "growth and contention")


def test_partitioned_add_mers_and_sharded():
    rng = np.random.default_rng(5)
    k, bits = 16, 4
    mers = rng.integers(0, 4 ** k, size=300000, dtype=np.uint64)
    mers[::2] = mers[1::2]
    hq = (rng.random(mers.size) < 0.5).astype(np.uint8)
    um, uv = dict_model_np(mers, hq, bits)
    h = qa.HashWithQuality(1 << 18, 2 * k, bits)
    h.set_mode("partition")
    assert h.add(mers, hq)
    m, v = sorted_pairs(*h.dump())
    assert (m == um).all() and (v == uv).all()
    # sharded handles in partition mode keep exactly their own keys
    parts = []
    for s in range(4):
        hs = qa.HashWithQuality(1 << 18, 2 * k, bits, shard_bits=2, shard_id=s)
        hs.set_mode("partition")
        assert hs.add(mers, hq)
        st = hs.done()
        assert st["instances"] + st["foreign"] == mers.size
        parts.append(hs.dump())
    m = np.concatenate([p[0] for p in parts])
    v = np.concatenate([p[1] for p in parts])
    assert np.unique(m).size == m.size
    m, v = sorted_pairs(m, v)
    assert (m == um).all() and (v == uv).all()


def test_modes_agree_at_full_size():
    """BASELINE config 2 at full size: direct and partitioned builds give the identical table"""
    k, bits, thresh = 24, 7, 38
    g = synth.make_genome(5_000_000)
    fq = synth.make_fastq(g, 2_000_000, 100)
    chunk = torch.from_numpy(fq).to("cuda:0")
    dumps = []
    for mode in ("direct", "partition"):
        h = qa.HashWithQuality(50_000_000, 2 * k, bits, min_qual=thresh)
        h.set_mode(mode)
        h.add_fastq_device(chunk.data_ptr(), chunk.numel())
        st = h.done()
        dumps.append((st, sorted_pairs(*h.dump())))
        h.close()
    (s0, (m0, v0)), (s1, (m1, v1)) = dumps
    assert s0["instances"] == s1["instances"] and s0["distinct"] == s1["distinct"] == m0.size
    assert (m0 == m1).all() and (v0 == v1).all()


@pytest.mark.parametrize("k", [5, 9, 11, 12, 13])
def test_short_k_quality_strings_that_look_like_bases(k):
    """windows of fewer than 12 bases can lie wholly behind a read's newline, inside its quality string: with
    qualities written in the letters A, C, G, T (all valid, high-quality characters) such a window must not
    become a k-mer. Read lengths are ragged so that last segments hold 1..12 k-mers."""
    rng = np.random.default_rng(500 + k)
    recs = []
    for i in range(3000):
        n = int(rng.integers(k, k + 40))
        seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n))
        qual = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n))
        recs.append(b"@read%d\n" % i + seq + b"\n+\n" + qual + b"\n")
    data = b"".join(recs)
    db = po.OracleDB(1 << 20, k, 7)
    n = db.add_fastq(data, 38, threads=2)
    om, ov = db.sorted_dump()
    h, st, m, v = build(data, k, 7, 1 << 20, "partition")
    assert st["instances"] == n and st["distinct"] == om.size
    assert (m == om).all() and (v == ov).all()
    h.close()
