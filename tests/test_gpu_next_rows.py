"""GPU parity tests of SURVEY 8f rows 2 and 3: the contaminant set built from FASTA with its
membership test fused into the lookup, and the Poisson cutoff, each against the CPU oracle."""
import os

import numpy as np
import pytest

import pyoracle as po
import quorum_amd as qa
from helpers import random_fastq, sorted_pairs, revcomp

pytestmark = pytest.mark.gpu

ADAPTERS = os.path.join(os.path.dirname(__file__), "golden", "adapter.fa")


@pytest.mark.parametrize("k", [24, 31, 15])
def test_contaminant_set_matches_oracle(k):
    data = open(ADAPTERS, "rb").read()
    db = po.OracleDB(1 << 16, k, 1)
    db.add_fasta(data)
    om, ov = db.sorted_dump()
    c = qa.contaminant_database(ADAPTERS, k)
    m, v = sorted_pairs(*c.dump())
    assert (m == om).all() and (v == ov).all()
    # membership: every key, every key with one base changed, random mers
    rng = np.random.default_rng(k)
    mut = om ^ (np.uint64(1) << rng.integers(0, 2 * k, size=om.size).astype(np.uint64))
    mut = np.array([min(x, revcomp(x, k)) for x in mut.tolist()], dtype=np.uint64)
    rnd = rng.integers(0, 4 ** k, size=20000, dtype=np.uint64)
    probe = np.concatenate([om, mut, rnd])
    want = db.lookup(probe) != 0
    got = c.is_member(probe)
    assert (got == want).all()
    assert want[:om.size].all() and not want[om.size:].all()
    c.close()


def test_add_fasta_multiline_and_errors():
    h = qa.HashWithQuality(1 << 10, 8, 1)
    h.add_fasta(b">a desc\nACGTA\nCGNAC\nGTT\n>b\nAC\n>c\nTTTTT")
    h.done()
    db = po.OracleDB(1 << 10, 4, 1)
    db.add_fasta(b">a desc\nACGTA\nCGNAC\nGTT\n>b\nAC\n>c\nTTTTT")
    m, v = sorted_pairs(*h.dump())
    om, ov = db.sorted_dump()
    assert (m == om).all() and (v == ov).all()
    with pytest.raises(qa.QhError, match="Unsupported format"):
        h.add_fasta(b"ACGT\n")
    h.add_fasta(b"")  # empty input is a no-op
    h.add_fasta(b">only a header\n")
    h.done()
    assert (sorted_pairs(*h.dump())[0] == om).all()
    with pytest.raises(RuntimeError, match="Failed to open contaminant file"):
        qa.contaminant_database("/nonexistent/adapter.fa", 24)


def test_lookup_fused_with_contaminant(tmp_path):
    """reads of which some carry a spliced adapter (SURVEY config C5's shape): one pass returns
    the database value and the contaminant flag of every canonical mer of the reads"""
    k, bits = 24, 7
    rng = np.random.default_rng(9)
    adapters = [b"".join(r.split(b"\n")[1:]) for r in open(ADAPTERS, "rb").read().split(b">")[1:]]
    data = bytearray(random_fastq(rng, 3000, max_len=150, k=k, genome=rng.integers(0, 4, size=50000)))
    recs = bytes(data).split(b"\n")
    for i in range(0, len(recs) - 3, 4 * 17):  # splice an adapter at the 3' end of every 17th read
        a = adapters[int(rng.integers(0, len(adapters)))]
        seq = recs[i + 1]
        if len(seq) > len(a):
            recs[i + 1] = seq[:len(seq) - len(a)] + a
    data = b"\n".join(recs)
    db = po.OracleDB(1 << 20, k, bits)
    db.add_fastq(data, 38)
    cdb = po.OracleDB(1 << 16, k, 1)
    cdb.add_fasta(open(ADAPTERS, "rb").read())
    h = qa.HashWithQuality(1 << 20, 2 * k, bits, min_qual=38)
    h.add_fastq(data)
    h.done()
    q = h.as_query()
    c = qa.contaminant_database(ADAPTERS, k)
    om, _ = db.sorted_dump()
    probe = np.concatenate([om, rng.integers(0, 4 ** k, size=5000, dtype=np.uint64)])
    vals, flags = q.lookup_with_contaminant(c, probe)
    assert (vals == db.lookup(probe)).all()
    want = cdb.lookup(probe) != 0
    assert (flags == want).all()
    assert 0 < want.sum() < om.size
    # mismatched mer lengths are refused like error_correct_reads.cc:703-705
    c15 = qa.contaminant_database(ADAPTERS, 15)
    with pytest.raises(qa.QhError, match="different than correction mer length"):
        q.lookup_with_contaminant(c15, probe[:4])
    c15.close(); c.close(); q.close()


def test_poisson_cutoff_matches_oracle():
    k, bits = 17, 7
    rng = np.random.default_rng(23)
    for genome_len, n_reads in [(3000, 2500), (20000, 1500), (200000, 800)]:
        data = random_fastq(rng, n_reads, max_len=140, k=k, genome=rng.integers(0, 4, size=genome_len))
        db = po.OracleDB(1 << 19, k, bits)
        db.add_fastq(data, 38)
        h = qa.HashWithQuality(1 << 19, 2 * k, bits, min_qual=38)
        h.add_fastq(data)
        h.done()
        q = h.as_query()
        assert q.stats() == db.stats()
        for cp, pt in [(0.01 / 3, 1e-6 / 0.01), (0.05 / 3, 1e-8), (0.5, 0.0), (1e-3, 0.5)]:
            assert q.poisson_cutoff(cp, pt) == db.poisson_cutoff(cp, pt), (genome_len, cp, pt)
        q.close()
    for lam, i in [(0.1, 3), (2.5, 10), (2.5, 11), (40.0, 60)]:
        assert qa.poisson_term(lam, i) == po.poisson_term(lam, i)
