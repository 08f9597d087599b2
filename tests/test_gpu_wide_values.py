"""GPU parity tests of the wide form (qh_spec.h): geometries whose remainder + probe + value fields do not fit one
64-bit slot keep the bits+1-wide values in a side array, like upstream's separate val_array
(src/mer_database.hpp:78-83) -- upstream takes every -b in 1..63 at any -m / -s (src/create_database.cc:111-112).
Build, growth, write, reopen, every lookup, the reductions and the corrector against the oracle, bit-exact."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import pyoracle as po
import quorum_amd as qa
from quorum_amd import _lib, synth
from helpers import random_fastq, sorted_pairs

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "quorum_amd", "bin")


def header_of(path):
    info = _lib.QhDbInfo()
    assert _lib.lib.qh_db_header_info(str(path).encode(), C.byref(info)) == 0
    return info


def file_values(path, info, bits):
    """(slot words, value fields) of a database file: vals image in atomic_bits_array layout"""
    raw = np.fromfile(path, dtype=np.uint64, offset=info.offset)
    slots, vimg = raw[:info.size], raw[info.size:]
    vb = bits + 1
    vpw = 64 // vb
    idx = np.arange(info.size, dtype=np.uint64)
    mask = np.uint64((1 << vb) - 1) if vb < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    fields = (vimg[idx // np.uint64(vpw)] >> ((idx % np.uint64(vpw)) * np.uint64(vb))) & mask
    assert vimg.size == (info.size + vpw - 1) // vpw
    return slots, fields


# (k, bits, table size): each is refused by a one-word slot (remainder + 7 probe bits + bits + 1 > 64)
WIDE = [(31, 32, 1 << 20), (31, 63, 1 << 20), (31, 14, 1 << 20), (24, 40, 1 << 16), (31, 21, 1 << 22), (17, 63, 1 << 12)]


@pytest.mark.parametrize("k,bits,size", WIDE)
def test_wide_build_write_reopen_matches_oracle(k, bits, size, tmp_path):
    rng = np.random.default_rng(100 + bits)
    data = random_fastq(rng, 2500, k=k)
    db = po.OracleDB(size, k, bits)
    n = db.add_fastq(data, 38)
    om, ov = db.sorted_dump()
    h = qa.HashWithQuality(size, 2 * k, bits, min_qual=38)
    h.add_fastq(data)
    st = h.done()
    m, v = sorted_pairs(*h.dump())
    assert st["instances"] == n and st["distinct"] == om.size
    assert m.size == om.size and (m == om).all() and (v == ov).all()
    # the same through the sequential parser, in pieces, into a table that has to double on the way
    h2 = qa.HashWithQuality(64, 2 * k, bits, min_qual=38)
    recs = data.split(b"@read")[1:]
    for i in range(0, len(recs), 700):
        h2.add_fastq_host_parsed(b"".join(b"@read" + r for r in recs[i:i + 700]), final=True)
    st2 = h2.done()
    m2, v2 = sorted_pairs(*h2.dump())
    assert st2["grows"] >= 3 and st2["instances"] == n and (m2 == om).all() and (v2 == ov).all()
    h2.close()
    # live query, then the file
    path = str(tmp_path / "wide.db")
    h.write(path, argv=["test"])
    info = header_of(path)
    assert info.bits == bits and info.key_len == 2 * k                                       # test_mer_database.cc:113-114
    assert info.offset % 8 == 0 and os.path.getsize(path) == info.offset + info.key_bytes + info.value_bytes  # :102-106
    slots, fields = file_values(path, info, bits)
    assert ((slots != 0) == (fields != 0)).all()
    assert np.array_equal(np.sort(fields[fields != 0]), np.sort(ov))
    absent = rng.integers(0, 1 << (2 * k), size=2000, dtype=np.uint64)
    probe = np.concatenate([om, absent])
    want = db.lookup(probe)
    for q in (h.as_query(), qa.DatabaseQuery(path)):
        assert q.header()["bits"] == bits and q.header()["key_len"] == 2 * k
        assert (q.lookup(probe) == want).all()
        assert (q.get_val(probe) == np.where(want & np.uint64(1), want >> np.uint64(1), np.uint64(0))).all()
        dm, dv = sorted_pairs(*q.dump())
        assert (dm == om).all() and (dv == ov).all()
        assert q.stats() == db.stats()
        assert (q.histo() == db.histo()).all()
        fm = om[:400] if k > 1 else om
        for bw in (0, 1):
            nk, cnt, uc, lv = q.get_best_alternatives(fm, bw)
            for i in range(0, fm.size, 7):
                on, oc, ou, ol = db.best_alternatives(int(fm[i]), bw)
                assert (int(nk[i]), cnt[i].tolist(), int(lv[i])) == (on, oc, ol)
                if on:
                    assert int(uc[i]) == ou
        q.close()
    h.close()


def test_wide_saturation_and_update_rule():
    """-b 14 beside a 31-mer in a 2^20-slot table: 43 remainder + 7 probe + 15 value bits. Counts saturate at 2^14 - 1
    (mer_database.hpp:83,107), high quality wipes low (:105), low after high is ignored (:107)."""
    k, bits = 31, 14
    mx = (1 << bits) - 1
    poly = b"@p\n" + b"A" * 20030 + b"\n+\n" + b"I" * 20030 + b"\n"      # 20000 high-quality AAAA...A
    lowq = b"@l\n" + b"C" * 600 + b"\n+\n" + b"#" * 600 + b"\n"          # 570 low-quality CCCC...C
    both = b"@b\n" + b"ACGT" * 20 + b"\n+\n" + b"#" * 80 + b"\n" + b"@b2\n" + b"ACGT" * 20 + b"\n+\n" + b"I" * 80 + b"\n"
    data = poly + lowq + both + lowq
    h = qa.HashWithQuality(1 << 20, 2 * k, bits, min_qual=38)
    h.add_fastq(data)
    h.done()
    db = po.OracleDB(1 << 20, k, bits)
    db.add_fastq(data, 38)
    om, ov = db.sorted_dump()
    m, v = sorted_pairs(*h.dump())
    assert (m == om).all() and (v == ov).all()
    got = dict(zip(m.tolist(), v.tolist()))
    assert got[0] == (mx << 1) | 1                                       # poly-A saturated, high quality
    assert got[int("01" * k, 2)] == (1140 << 1) | 0                      # poly-C (canonical of itself vs poly-G), low quality
    # mers as given, all four orders of the update rule, beyond 2^14 sightings
    h2 = qa.HashWithQuality(1 << 20, 2 * k, bits)
    a, b, c = np.uint64(12345678901234567), np.uint64(7), np.uint64(99999999999)
    mers = np.concatenate([np.full(mx + 10, a), np.full(5, b), np.full(3, b), np.full(4, c), np.full(2, c)])
    hq = np.concatenate([np.zeros(mx + 10), np.zeros(5), np.ones(3), np.ones(4), np.zeros(2)]).astype(np.uint8)
    assert h2.add(mers, hq)
    got = dict(zip(*[x.tolist() for x in h2.dump()]))
    assert got == {int(a): mx << 1, int(b): (3 << 1) | 1, int(c): (4 << 1) | 1}
    h.close(); h2.close()


def test_wide_counts_above_32_bits(tmp_path):
    """No input here reaches 2^32 sightings of one mer, so the counts a -b 40 / -b 63 database may hold are put into
    the file's values image by hand (atomic_bits_array: one field per word from 33 value bits on) and every reader must
    return them whole: lookups, get_val, best alternatives, iteration, the histogram's last bin, the Poisson sums."""
    k = 31
    rng = np.random.default_rng(5)
    data = random_fastq(rng, 800, k=k)
    for bits in (40, 63):
        h = qa.HashWithQuality(1 << 18, 2 * k, bits, min_qual=38)
        h.add_fastq(data)
        h.done()
        path = str(tmp_path / ("big%d.db" % bits))
        h.write(path)
        h.close()
        info = header_of(path)
        slots, fields = file_values(path, info, bits)
        occ = np.flatnonzero(slots)
        pick = occ[:: max(1, occ.size // 50)][:40]
        big = ((np.uint64(1) << np.uint64(bits)) - np.uint64(1)) - np.arange(pick.size, dtype=np.uint64) * np.uint64(977)
        newv = (big << np.uint64(1)) | (np.arange(pick.size, dtype=np.uint64) & np.uint64(1))
        assert int(big.min()) > (1 << 32)
        with open(path, "r+b") as f:
            for s, nv in zip(pick.tolist(), newv.tolist()):
                f.seek(info.offset + info.key_bytes + 8 * s)                # vpw = 1
                f.write(np.uint64(nv).tobytes())
        q = qa.DatabaseQuery(path)
        dm, dv = q.dump()
        by_val = dict(zip(dv.tolist(), dm.tolist()))
        mers = np.array([by_val[int(x)] for x in newv.tolist()], dtype=np.uint64)
        assert (q.lookup(mers) == newv).all()
        assert (q.get_val(mers) == np.where(newv & np.uint64(1), newv >> np.uint64(1), np.uint64(0))).all()
        d, t = q.stats()
        hqv = dv[(dv & np.uint64(1)) == 1]
        assert d == hqv.size and t == int(sum(int(x) >> 1 for x in hqv.tolist())) % (1 << 64)
        hist = q.histo()
        c = np.minimum(dv >> np.uint64(1), np.uint64(1000))
        for lvl in (0, 1):
            assert int(hist[1000, lvl]) == int(((c == 1000) & ((dv & np.uint64(1)) == lvl)).sum()) >= 19
        # the four siblings of a patched mer: its own count comes back whole
        nk, cnt, uc, lv = q.get_best_alternatives(mers, 0)
        for i in range(mers.size):
            own = int(mers[i]) & 3
            if int(newv[i]) & 1 or int(lv[i]) == 0:
                assert int(cnt[i][own]) == int(newv[i]) >> 1
        q.close()


def test_wide_corrector_matches_oracle():
    """the corrector on a database in the wide form (word mode 3 of qh_correct.hip): both text streams, byte for byte"""
    k, bits = 31, 33
    g = synth.make_genome(20000, seed=3)
    fq = synth.make_fastq(g, 6000, 100, err=0.01, seed=4, tails=True).tobytes()
    odb = po.OracleDB(1 << 19, k, bits)
    odb.add_fastq(fq, 38)
    h = qa.HashWithQuality(1 << 19, 2 * k, bits, min_qual=38)
    h.add_fastq(fq)
    h.done()
    q = h.as_query()
    cutoff = odb.poisson_cutoff(0.01 / 3, 1e-6 / 0.01)
    for kw in ({}, {"homo_trim": 3, "no_discard": True}):
        ec = qa.ErrorCorrector(q, **kw)
        assert ec.cutoff == cutoff
        fa, lg, st, used = ec.correct_fastq(fq)
        par = po.ec_params(cutoff, homo_trim=kw.get("homo_trim"), no_discard=kw.get("no_discard", False))
        ofa, olg, n = odb.correct_fastq(fq, par)
        assert used == len(fq) and fa == ofa and lg == olg and st["reads"] == n
        assert st["substitutions"] > 100
        ec.close()
    q.close(); h.close(); odb.close()


def test_wide_sharded_and_direct_items():
    """two shard handles of a wide geometry fed the same items: their union is the unsharded table; the sharded file
    opens as one database"""
    k, bits = 31, 36
    rng = np.random.default_rng(77)
    mers = rng.integers(0, 1 << 62, size=60000, dtype=np.uint64)
    mers[::2] = mers[1::2]
    hq = (rng.random(mers.size) < 0.5).astype(np.uint8)
    one = qa.HashWithQuality(1 << 18, 2 * k, bits)
    assert one.add(mers, hq)
    wm, wv = sorted_pairs(*one.dump())
    parts = []
    shards = [qa.HashWithQuality(1 << 18, 2 * k, bits, shard_bits=1, shard_id=s) for s in (0, 1)]
    for s in shards:
        assert s.add(mers, hq)
        st = s.done()
        assert st["foreign"] > 0
        parts.append(s.dump())
    gm, gv = sorted_pairs(np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]))
    assert (gm == wm).all() and (gv == wv).all()
    for s in shards + [one]:
        s.close()


def test_cli_takes_reference_legal_bits(tmp_path):
    """quorum_create_database -m 31 -s 100M -b 32 (and -b 63): refused until round 3, legal upstream
    (src/create_database.cc:111-112). Built, written, read back by query_mer_database and the class mirror."""
    rng = np.random.default_rng(9)
    data = random_fastq(rng, 1500, k=31)
    fq = tmp_path / "r.fq"
    fq.write_bytes(data)
    for bits, size in ((32, "100M"), (63, "1M")):
        out = str(tmp_path / ("cli%d.db" % bits))
        r = subprocess.run([os.path.join(BIN, "quorum_create_database"), "-s", size, "-m", "31", "-b", str(bits), "-q", "38",
                            "-o", out, str(fq)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        db = po.OracleDB(1 << 20, 31, bits)
        db.add_fastq(data, 38)
        om, ov = db.sorted_dump()
        q = qa.DatabaseQuery(out)
        assert q.header()["bits"] == bits
        dm, dv = sorted_pairs(*q.dump())
        assert (dm == om).all() and (dv == ov).all()
        q.close()
        mer = qa.mer_to_str(int(om[0]), 31)
        r = subprocess.run([os.path.join(BIN, "query_mer_database"), out, mer], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert r.stdout.split()[-2:] == ["val:%d" % (int(ov[0]) >> 1), "qual:%d" % (int(ov[0]) & 1)]
