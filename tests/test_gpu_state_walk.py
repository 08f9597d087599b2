"""Random walks over the builder's states: chunks of every size through every entry point (host text, device text back to
back -- early preparation --, host-parsed text, mer lists on the host and on the device), strategies and deferral budgets
switched on the way, collections, window passes started without waiting, lookups on the live table, dumps and resets
(noted while work is in flight, applied at the next collection) in between, on tables small enough to grow.
After every observation the table must equal the oracle's, which received the same inputs one by one
(hash_with_quality::add is order independent, src/mer_database.hpp:94-113)."""
import os

import numpy as np
import pytest
import torch

import pyoracle as po
import quorum_amd as qa
from quorum_amd import synth
from helpers import sorted_pairs

pytestmark = pytest.mark.gpu

L = 100
SEEN = {"early_preps": 0, "grows": 0, "window_passes": 0, "walks": 0}   # what the walks reached (checked by the last test)


def note(st):
    for key in ("early_preps", "grows", "window_passes"):
        SEEN[key] += st[key]


def reads_for(k, seed):
    g = synth.make_genome(300_000, seed=seed)
    return synth.make_fastq(g, 260_000, L, err=0.004, n_rate=0.001, seed=seed + 1, tails=True), synth.record_bytes(L)


WALKS = [(1, 24, 7, 1 << 22), (2, 31, 3, 1 << 20), (3, 17, 5, 1 << 23), (4, 24, 1, 1 << 18), (5, 31, 12, 1 << 22),
         (6, 12, 7, 1 << 22)]
if os.environ.get("QH_WALK_EXTRA"):   # a soak run: QH_WALK_EXTRA=<n> adds n more walks over drawn geometries
    _r = np.random.default_rng(int(os.environ.get("QH_WALK_SEED", "2024")))
    for _i in range(int(os.environ["QH_WALK_EXTRA"])):
        WALKS.append((100 + _i, int(_r.choice([9, 12, 15, 17, 20, 24, 27, 31])), int(_r.choice([1, 3, 7, 12, 20])),
                      1 << int(_r.integers(14, 25))))


@pytest.mark.parametrize("seed,k,bits,slots", WALKS)
def test_random_walk_matches_oracle(seed, k, bits, slots):
    rng = np.random.default_rng(seed)
    fq, RB = reads_for(k, 100 + seed)
    n_reads = fq.size // RB
    Q = 38
    h = qa.HashWithQuality(slots, 2 * k, bits, min_qual=Q)
    db = po.OracleDB(1 << 24, k, bits)
    expect_inst = 0
    cursor = 0
    keep = []            # device tensors stay alive until the handle has collected their chunk

    def take(n):
        nonlocal cursor
        if cursor + n > n_reads:
            cursor = 0
        part = fq[cursor * RB:(cursor + n) * RB]
        cursor += n
        return part

    def check():
        om, ov = db.sorted_dump()
        m, v = sorted_pairs(*h.dump())
        st = h.stats()
        assert st["instances"] == expect_inst and st["distinct"] == om.size == m.size, (st, om.size)
        assert (m == om).all() and (v == ov).all()

    for step in range(60):
        op = rng.choice(["text", "dev", "dev", "dev", "parsed", "mers", "devmers", "sync", "lookup", "dump", "reset",
                         "defer", "mode", "flush"])
        if op in ("text", "dev", "parsed"):
            n = int(rng.choice([1, 40, 3_000, 25_000, 60_000]))
            part = take(n)
            expect_inst += db.add_fastq(part.tobytes(), Q, threads=8)
            if op == "text":
                h.add_fastq(part.tobytes())
            elif op == "parsed":
                h.add_fastq_host_parsed(part.tobytes(), final=True)
            else:
                t = torch.from_numpy(part.copy()).to("cuda:0")
                torch.cuda.synchronize()
                keep.append(t)
                h.add_fastq_device(t.data_ptr(), t.numel())
        elif op in ("mers", "devmers"):
            n = int(rng.choice([1, 500, 200_000, 900_000]))
            mers = rng.integers(0, 1 << (2 * k), size=n, dtype=np.uint64)
            mers[::3] = mers[0]                                # one hot key among them
            q = (rng.random(n) < 0.4).astype(np.uint8)
            db.add_mers(mers, q)
            expect_inst += n
            if op == "mers":
                assert h.add(mers, q)
            else:
                t = torch.from_numpy(((mers << np.uint64(1)) | q).view(np.int64)).to("cuda:0")
                torch.cuda.synchronize()
                keep.append(t)
                h.add_mers_device(t.data_ptr(), n)
        elif op == "sync":
            h.sync()
            keep.clear()
        elif op == "lookup":
            om, ov = db.sorted_dump()
            if om.size:
                pick = om[rng.integers(0, om.size, size=min(5000, om.size))]
                q = h.as_query()
                got = q.lookup(pick)
                q.close()
                assert (got == ov[np.searchsorted(om, pick)]).all()
            keep.clear()
        elif op == "dump":
            check()
            keep.clear()
        elif op == "reset":
            note(h.stats())
            h.reset()
            keep.clear()
            db = po.OracleDB(1 << 24, k, bits)
            expect_inst = 0
        elif op == "flush":
            h.flush()                       # the pass over the table starts; whatever comes next collects it
        elif op == "defer":
            h.set_defer(int(rng.choice([0, 1 << 27, 1 << 30, -1])))
        elif op == "mode":
            h.set_mode(str(rng.choice(["auto", "partition", "direct"])))
    note(h.done())
    check()
    h.close()
    SEEN["walks"] += 1


def test_the_walks_reached_the_states_they_are_for():
    assert SEEN["walks"] == len(WALKS), "run the whole file"
    assert SEEN["early_preps"] > 10 and SEEN["grows"] > 3 and SEEN["window_passes"] > 10, SEEN
