"""Seeded synthetic Illumina-like reads (SURVEY.md 8d / BASELINE.md): the inputs of
the benchmark and of the size-independent parity tests. Input generation only --
nothing here is on the measured path.

Genome: uniform i.i.d. ACGT. Reads: uniform start, uniform strand, substitutions at
rate `err` to one of the three other bases, 'N' at rate `n_rate`, Phred+33
qualities 'I' (40) with substituted bases at U[2,15]; optional low-quality tails
(per-read tail length U[0,40], quality decaying linearly from 40 to 2).
Every record is strict 4-line FASTQ of the same byte length:
    @r<9 digits>\\n<L bases>\\n+\\n<L quals>\\n       (2L + 16 bytes)
"""
import numpy as np

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_genome(n_bases, seed=42):
    """uint8 codes 0..3"""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 4, size=n_bases, dtype=np.uint8)


def record_bytes(read_len):
    return 2 * read_len + 16


def make_fastq(genome, n_reads, read_len, err=0.01, n_rate=0.0005, seed=42, tails=False, first_id=0,
               batch=200_000):
    """-> np.uint8 array holding n_reads FASTQ records"""
    L = read_len
    rb = record_bytes(L)
    out = np.empty((n_reads, rb), dtype=np.uint8)
    rng = np.random.default_rng([seed, first_id])
    G = genome.size
    ar = np.arange(L)
    pow10 = 10 ** np.arange(8, -1, -1, dtype=np.int64)
    for b0 in range(0, n_reads, batch):
        n = min(batch, n_reads - b0)
        starts = rng.integers(0, G - L + 1, size=n)
        codes = genome[starts[:, None] + ar[None, :]]
        rc = rng.random(n) < 0.5
        codes[rc] = 3 - codes[rc, ::-1]
        errs = rng.random((n, L)) < err
        ne = int(errs.sum())
        codes[errs] = (codes[errs] + rng.integers(1, 4, size=ne, dtype=np.uint8)) & 3
        qual = np.full((n, L), ord("I"), dtype=np.uint8)
        if tails:
            T = rng.integers(0, 41, size=n)
            j = ar[None, :] - (L - T[:, None])  # >= 0 inside the tail
            intail = j >= 0
            q = 40.0 - (j + 1) * 38.0 / np.maximum(T[:, None], 1)
            qual[intail] = (np.clip(np.rint(q), 2, 40).astype(np.uint8) + 33)[intail]
        qual[errs] = rng.integers(2, 16, size=ne, dtype=np.uint8) + 33
        bases = _BASES[codes]
        if n_rate > 0:
            bases[rng.random((n, L)) < n_rate] = ord("N")
        ids = first_id + b0 + np.arange(n, dtype=np.int64)
        rec = out[b0:b0 + n]
        rec[:, 0] = ord("@")
        rec[:, 1] = ord("r")
        rec[:, 2:11] = ((ids[:, None] // pow10[None, :]) % 10 + 48).astype(np.uint8)
        rec[:, 11] = 10
        rec[:, 12:12 + L] = bases
        rec[:, 12 + L] = 10
        rec[:, 13 + L] = ord("+")
        rec[:, 14 + L] = 10
        rec[:, 15 + L:15 + 2 * L] = qual
        rec[:, 15 + 2 * L] = 10
    return out.reshape(-1)


def expected_instances(n_reads, read_len, k):
    """upper bound on emitted k-mers (exact when there is no 'N')"""
    return n_reads * max(0, read_len - k + 1)
