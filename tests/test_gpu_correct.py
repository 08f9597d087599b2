"""GPU parity tests of SURVEY 8f row 1: the read corrector (quorum_error_correct_reads) on the
GPU against the CPU oracle (oracle/error_correct_oracle.c), both text streams byte for byte.

Parity unpinned: the reference holds no fixture for the corrector; the oracle is a restatement
of src/error_correct_reads.cc checked by hand-derived cases in tests/test_correct_oracle.py."""
import os

import numpy as np
import pytest

import pyoracle as po
import quorum_amd as qa
from quorum_amd import synth
from helpers import random_fastq

pytestmark = pytest.mark.gpu

ADAPTERS = os.path.join(os.path.dirname(__file__), "golden", "adapter.fa")


def first_diff(a, b):
    la, lb = a.split(b"\n"), b.split(b"\n")
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return "line %d:\n gpu    %r\n oracle %r" % (i, x[:300], y[:300])
    return "lengths %d vs %d lines" % (len(la), len(lb))


def build_pair(fq, k, bits=7, thresh=38, size=1 << 20):
    odb = po.OracleDB(size, k, bits)
    odb.add_fastq(fq, thresh)
    h = qa.HashWithQuality(size, 2 * k, bits, min_qual=thresh)
    h.add_fastq(fq)
    h.done()
    return odb, h, h.as_query()


def check_parity(odb, q, fq, ocont=None, gcont=None, via="host", **kw):
    okw = dict(kw)
    cutoff = okw.pop("cutoff", None)
    if cutoff is None:
        cutoff = odb.poisson_cutoff(0.01 / 3, 1e-6 / 0.01)
    ec = qa.ErrorCorrector(q, contaminant=gcont, cutoff=kw.get("cutoff"), **{a: b for a, b in kw.items() if a != "cutoff"})
    assert ec.cutoff == cutoff
    if via == "host":
        fa, lg, st, used = ec.correct_fastq(fq)
        assert used == len(fq)
    else:
        import torch
        t = torch.frombuffer(bytearray(fq), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        ec.correct_device(t.data_ptr(), len(fq))
        fa, lg, st = ec.fetch()
    par = po.ec_params(cutoff, skip=okw.get("skip", 1), good=okw.get("good", 2), anchor=okw.get("anchor_count", 3),
                       min_count=okw.get("min_count", 1), window=okw.get("window", 10), error=okw.get("error", 3),
                       qual_cutoff=(ord(okw["qual_cutoff_char"]) if "qual_cutoff_char" in okw
                                    else okw.get("qual_cutoff_value", 127)),
                       trim_contaminant=okw.get("trim_contaminant", False), homo_trim=okw.get("homo_trim"),
                       no_discard=okw.get("no_discard", False))
    ofa, olg, n = odb.correct_fastq(fq, par, contaminant=ocont)
    assert fa == ofa, first_diff(fa, ofa)
    assert lg == olg, first_diff(lg, olg)
    assert st["reads"] == n
    assert st["substitutions"] == ofa.count(b":sub:")
    assert st["truncations"] == ofa.count(b"_trunc")
    assert st["skipped"] == olg.count(b"Skipped ")
    ec.close()
    return fa, lg, st


@pytest.fixture(scope="module")
def bacterial():
    g = synth.make_genome(30000, seed=5)
    fq = synth.make_fastq(g, 12000, 100, err=0.01, seed=7, tails=True).tobytes()
    odb, h, q = build_pair(fq, 24)
    yield g, fq, odb, q
    q.close(); h.close(); odb.close()


@pytest.mark.parametrize("via", ["host", "device"])
def test_default_switches(bacterial, via):
    g, fq, odb, q = bacterial
    fa, lg, st = check_parity(odb, q, fq, via=via)
    assert st["corrected"] > 11000 and st["substitutions"] > 5000


@pytest.mark.parametrize("load", ["0", "0.3", "0.05"])
def test_sparser_private_table(bacterial, load, monkeypatch):
    """a database at load 0.41: the corrector re-hashes it into a table of load <= QH_EC_TABLE_LOAD for its walks (0: works
    on the database's own table; 0.05: three doublings) -- the text is the oracle's either way"""
    g, fq, _, _ = bacterial
    odb, h, q = build_pair(fq, 24, size=1 << 19)
    assert odb.distinct() > 0.4 * (1 << 19) and h.stats()["slots"] == 1 << 19
    monkeypatch.setenv("QH_EC_TABLE_LOAD", load)
    check_parity(odb, q, fq, via="device")
    assert (q.lookup(np.arange(1000, dtype=np.uint64)) == odb.lookup(np.arange(1000, dtype=np.uint64))).all()  # the handle is untouched
    q.close(); h.close(); odb.close()


@pytest.mark.parametrize("kw", [
    dict(window=0, error=0),                     # window = k, error = k / 2 (error_correct_reads.cc:206-207)
    dict(skip=0, good=1, anchor_count=1),
    dict(skip=5, good=4, anchor_count=6, min_count=2),
    dict(cutoff=2), dict(cutoff=9),
    dict(qual_cutoff_value=53), dict(qual_cutoff_char="5"),
    dict(homo_trim=2), dict(homo_trim=-3), dict(homo_trim=40),
    dict(no_discard=True, anchor_count=60),      # nearly every read is skipped
    dict(window=4, error=1), dict(window=30, error=2),
    dict(poisson_threshold=1e-2), dict(apriori_error_rate=0.05),
])
def test_switches(bacterial, kw):
    g, fq, odb, q = bacterial
    par = dict(kw)
    # the oracle helper takes the raw probabilities
    apr = par.pop("apriori_error_rate", 0.01)
    pth = par.pop("poisson_threshold", 1e-6)
    cutoff = par.get("cutoff")
    if cutoff is None:
        cutoff = odb.poisson_cutoff(apr / 3, pth / apr)
    ec = qa.ErrorCorrector(q, apriori_error_rate=apr, poisson_threshold=pth, **par)
    assert ec.cutoff == cutoff
    fa, lg, st, used = ec.correct_fastq(fq)
    opar = po.ec_params(cutoff, skip=par.get("skip", 1), good=par.get("good", 2), anchor=par.get("anchor_count", 3),
                        min_count=par.get("min_count", 1), window=par.get("window", 10), error=par.get("error", 3),
                        qual_cutoff=(ord(par["qual_cutoff_char"]) if "qual_cutoff_char" in par
                                     else par.get("qual_cutoff_value", 127)),
                        homo_trim=par.get("homo_trim"), no_discard=par.get("no_discard", False),
                        apriori_error_rate=apr, poisson_threshold=pth)
    ofa, olg, n = odb.correct_fastq(fq, opar)
    assert fa == ofa, first_diff(fa, ofa)
    assert lg == olg, first_diff(lg, olg)
    ec.close()


@pytest.mark.parametrize("k,bits", [(15, 7), (31, 7), (24, 3), (20, 12), (12, 34), (10, 7)])  # (12, 34): values wider than 32 bits; (10, 7): no sibling walk
def test_other_geometries(k, bits):
    g = synth.make_genome(20000, seed=k)
    fq = synth.make_fastq(g, 6000, 120, err=0.02, seed=k + 1, tails=True).tobytes()
    odb, h, q = build_pair(fq, k, bits=bits)
    check_parity(odb, q, fq)
    check_parity(odb, q, fq, via="device", window=0, error=0)
    q.close(); h.close(); odb.close()


def test_long_reads_text_written_unstaged():
    """256 reads of 400 bp do not fit the format kernel's LDS staging: the direct-write path"""
    g = synth.make_genome(30000, seed=31)
    fq = synth.make_fastq(g, 3000, 400, err=0.01, seed=32, tails=True).tobytes()
    odb, h, q = build_pair(fq, 24)
    check_parity(odb, q, fq, via="device")
    check_parity(odb, q, fq, no_discard=True, anchor_count=50)
    q.close(); h.close(); odb.close()


def test_ragged_reads_and_odd_characters():
    """N, IUPAC, lowercase, reads around and below k, negative quality bytes, long headers"""
    rng = np.random.default_rng(11)
    genome = rng.integers(0, 4, size=4000)
    fq = random_fastq(rng, 6000, max_len=160, genome=genome, k=24)
    odb, h, q = build_pair(fq, 24)
    check_parity(odb, q, fq)
    check_parity(odb, q, fq, via="device", no_discard=True)
    check_parity(odb, q, fq, skip=0, good=1, anchor_count=1, min_count=0, homo_trim=1)
    q.close(); h.close(); odb.close()


@pytest.mark.parametrize("trim", [False, True])
def test_contaminant(trim):
    """reads with adapters spliced in: discarded ("Contaminated read") or trimmed (--trim-contaminant)"""
    k = 24
    rng = np.random.default_rng(3)
    g = synth.make_genome(20000, seed=9)
    fq = synth.make_fastq(g, 6000, 100, err=0.01, seed=10).tobytes()
    adapters = [l.strip() for l in open(ADAPTERS, "rb").read().split(b"\n") if l and not l.startswith(b">")]
    adapters = [a for a in adapters if len(a) >= k]
    recs = fq.split(b"\n")
    for i in range(1, len(recs), 4):
        if rng.random() < 0.05:
            a = adapters[int(rng.integers(0, len(adapters)))]
            at = int(rng.integers(0, 100 - k))
            s = bytearray(recs[i])
            n = min(len(a), 100 - at)
            s[at:at + n] = a[:n]
            recs[i] = bytes(s)
    fq = b"\n".join(recs)
    odb, h, q = build_pair(fq, k)
    oc = po.OracleDB(1 << 16, k, 1)
    oc.add_fasta(open(ADAPTERS, "rb").read())
    gc = qa.contaminant_database(ADAPTERS, k)
    fa, lg, st = check_parity(odb, q, fq, ocont=oc, gcont=gc, trim_contaminant=trim)
    if trim:
        assert b"Contaminated" not in lg and st["truncations"] > 100
    else:
        assert lg.count(b"Contaminated read") > 100
    gc.close(); q.close(); h.close(); odb.close(); oc.close()


@pytest.mark.parametrize("k", [24, 9])
def test_contaminant_bit_filter_changes_nothing(k, monkeypatch):
    """the bit filter in front of the contaminant table (a bit per hash of a contaminant k-mer; 2^22 bits, fewer for
    short mers) only ever answers "not a contaminant" for mers that are none: same text with and without it, and both
    equal to the oracle's; at k = 9 the read's own k-mers collide with the adapters' all the time"""
    rng = np.random.default_rng(13)
    g = synth.make_genome(20000, seed=19)
    fq = synth.make_fastq(g, 5000, 100, err=0.01, seed=20).tobytes()
    adapters = [l.strip() for l in open(ADAPTERS, "rb").read().split(b"\n") if l and not l.startswith(b">")]
    adapters = [a for a in adapters if len(a) >= 24]
    recs = fq.split(b"\n")
    for i in range(1, len(recs), 4):
        if rng.random() < 0.05:
            a = adapters[int(rng.integers(0, len(adapters)))]
            at = int(rng.integers(0, 70))
            s = bytearray(recs[i])
            n = min(len(a), 100 - at)
            s[at:at + n] = a[:n]
            recs[i] = bytes(s)
    fq = b"\n".join(recs)
    odb, h, q = build_pair(fq, k)
    oc = po.OracleDB(1 << 16, k, 1)
    oc.add_fasta(open(ADAPTERS, "rb").read())
    gc = qa.contaminant_database(ADAPTERS, k)
    fa1, lg1, st1 = check_parity(odb, q, fq, ocont=oc, gcont=gc, trim_contaminant=True, cutoff=4)
    monkeypatch.setenv("QH_EC_NO_CONTAMINANT_BITS", "1")
    fa2, lg2, st2 = check_parity(odb, q, fq, ocont=oc, gcont=gc, trim_contaminant=True, cutoff=4)
    assert fa1 == fa2 and lg1 == lg2 and st1 == st2 and st1["truncations"] > 100
    gc.close(); q.close(); h.close(); odb.close(); oc.close()


def test_config5_shape_k31_paired_tails_adapters():
    """BASELINE configs[4]: 150 bp reads, 3 % errors, low-quality tails, k = 31, adapters spliced at the 3' end
    of 1 % of the reads, contaminant set = data/adapter.fa, --trim-contaminant (64-bit tags: the wide kernel)"""
    k = 31
    rng = np.random.default_rng(55)
    g = synth.make_genome(40000, seed=51)
    fq = synth.make_fastq(g, 10000, 150, err=0.03, seed=52, tails=True).tobytes()
    adapters = [l.strip() for l in open(ADAPTERS, "rb").read().split(b"\n") if l and not l.startswith(b">")]
    adapters = [a for a in adapters if len(a) >= k]
    recs = fq.split(b"\n")
    for i in range(1, len(recs), 4):
        if rng.random() < 0.01:
            a = adapters[int(rng.integers(0, len(adapters)))]
            n = min(len(a), 60)
            recs[i] = recs[i][:150 - n] + a[:n]
    fq = b"\n".join(recs)
    odb, h, q = build_pair(fq, k, size=1 << 22)
    oc = po.OracleDB(1 << 16, k, 1)
    oc.add_fasta(open(ADAPTERS, "rb").read())
    gc = qa.contaminant_database(ADAPTERS, k)
    fa, lg, st = check_parity(odb, q, fq, ocont=oc, gcont=gc, trim_contaminant=True, via="device")
    assert st["truncations"] > 50 and st["substitutions"] > 10000
    check_parity(odb, q, fq, ocont=oc, gcont=gc, trim_contaminant=False, homo_trim=3, no_discard=True)
    gc.close(); q.close(); h.close(); odb.close(); oc.close()


def test_multiline_records_take_the_host_parser():
    g = synth.make_genome(8000, seed=2)
    fq = synth.make_fastq(g, 1500, 100, err=0.01, seed=4).tobytes()
    odb, h, q = build_pair(fq, 24)
    lines = fq.split(b"\n")
    out = []
    for i in range(0, len(lines) - 1, 4):
        hd, s, pl, ql = lines[i:i + 4]
        if (i // 4) % 3 == 0:
            out += [hd + b" multi", s[:37], s[37:80], s[80:], pl + hd[1:], ql[:50], ql[50:]]
        else:
            out += [hd, s, pl, ql]
    ml = b"\n".join(out) + b"\n"
    check_parity(odb, q, ml)
    # a chunk that ends inside a record: the complete records are corrected, the tail is left
    ec = qa.ErrorCorrector(q)
    cut = len(ml) - 57
    fa, lg, st, used = ec.correct_fastq(ml[:cut], final=False)
    assert 0 < used < cut and ml[used:used + 1] == b"@"
    ofa, olg, n = odb.correct_fastq(ml[:used], po.ec_params(ec.cutoff))
    assert fa == ofa and lg == olg and st["reads"] == n
    ec.close(); q.close(); h.close(); odb.close()


def test_errors_and_empty_input(bacterial):
    g, fq, odb, q = bacterial
    ec = qa.ErrorCorrector(q)
    fa, lg, st, used = ec.correct_fastq(b"")
    assert fa == b"" and lg == b"" and used == 0 and st["reads"] == 0
    with pytest.raises(qa.QhError, match="Unsupported format"):
        ec.correct_fastq(b"ACGT\nACGT\n")
    with pytest.raises(qa.QhError, match="wrong number of quals"):
        ec.correct_fastq(b"@r\nACGT\n+\nIII\n@s\nAC\n+\nII\n")
    with pytest.raises(qa.QhError, match="Truncated"):
        ec.correct_fastq(b"@r\nACGT\n")
    # the handle is still usable after an error
    fa, lg, st, used = ec.correct_fastq(fq[:2160])
    assert st["reads"] == 10
    ec.close()
    with pytest.raises(ValueError):
        qa.ErrorCorrector(q, qual_cutoff_char="ab")
    with pytest.raises(ValueError):
        qa.ErrorCorrector(q, qual_cutoff_value=200)
    c15 = qa.contaminant_database(ADAPTERS, 15)
    with pytest.raises(qa.QhError, match="Contaminant mer length"):
        qa.ErrorCorrector(q, contaminant=c15)
    c15.close()


def test_corrected_reads_come_from_the_genome(bacterial):
    """size-independent property: with 40x coverage and 1 % errors nearly every corrected read is
    an exact substring of the genome (either strand)"""
    g, fq, odb, q = bacterial
    ec = qa.ErrorCorrector(q)
    fa, lg, st, used = ec.correct_fastq(fq)
    gs = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[g])
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    grc = gs.translate(comp)[::-1]
    lines = fa.split(b"\n")
    seqs = lines[1::2]
    ok = sum(1 for s in seqs if s and (s in gs or s in grc))
    raw = fq.split(b"\n")[1::4]
    raw_ok = sum(1 for s in raw if s in gs or s in grc)
    assert ok > 0.97 * st["corrected"] and raw_ok < 0.5 * len(raw)
    ec.close()


def test_read_of_2pow24_bases_is_rejected():
    """the error log keeps positions in 24 bits: a read of 2^24 bases or more must reject the chunk (QH_EFORMAT) on the
    device-indexed path and on the host-parsed path instead of corrupting its log"""
    import torch
    from quorum_amd import _lib
    rng = np.random.default_rng(3)
    small = random_fastq(rng, 200, k=24)
    odb, h, q = build_pair(small, 24)
    n = (1 << 24) + 5
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].tobytes()
    big = b"@long\n" + seq + b"\n+\n" + b"I" * n + b"\n"
    ec = qa.ErrorCorrector(q, cutoff=4)
    with pytest.raises(_lib.QhError) as e:      # fast path rejects on the GPU, the host parser then names the read
        ec.correct_fastq(small + big)
    assert e.value.status == _lib.QH_EFORMAT and "2^24" in str(e.value)
    t = torch.frombuffer(bytearray(small + big), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    ec.correct_device(t.data_ptr(), t.numel())
    with pytest.raises(_lib.QhError) as e:
        ec.fetch()
    assert e.value.status == _lib.QH_EFORMAT and "2^24" in str(e.value)
    # one base fewer is fine
    ok = b"@long\n" + seq[:(1 << 24) - 1] + b"\n+\n" + b"I" * ((1 << 24) - 1) + b"\n"
    fa, lg, st, used = ec.correct_fastq(small + ok)
    assert used == len(small + ok) and st["reads"] == 201
    ec.close()
