"""Shared test helpers: an independent dictionary model of the reference semantics
(final value = (min(H,max),1) if H>0 else (min(L,max),0)), a
window-based extractor written differently from both the oracle's rolling loop and
the GPU's bit-plane windows, and random FASTQ with the edge cases the path has."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CODE = {"A": 0, "C": 1, "G": 2, "T": 3, "a": 0, "c": 1, "g": 2, "t": 3}


def golden_cases():
    with open(os.path.join(HERE, "golden", "known_answers.json")) as f:
        cases = json.load(f)
    for c in cases:
        c["data"] = bytes.fromhex(c["fastq_hex"]) if "fastq_hex" in c else c["fastq"].encode()
    return cases


def mer_from_str(s):
    m = 0
    for ch in s:
        m = (m << 2) | CODE[ch]
    return m


def revcomp(m, k):
    r = 0
    for _ in range(k):
        r = (r << 2) | (3 - (m & 3))
        m >>= 2
    return r


def window_kmers(seq, qual, k, thresh):
    """every window of k consecutive ACGT bases -> (canonical, hq); hq iff every
    quality of the window is >= thresh as signed chars"""
    out = []
    t = thresh - 256 if thresh >= 128 else thresh
    for i in range(len(seq) - k + 1):
        w = seq[i:i + k]
        try:
            m = 0
            for ch in w:
                m = (m << 2) | CODE[chr(ch)]
        except KeyError:
            continue
        r = revcomp(m, k)
        hq = all((q - 256 if q >= 128 else q) >= t for q in qual[i:i + k])
        out.append((min(m, r), 1 if hq else 0))
    return out


def dict_model(instances, bits):
    """[(mer, hq)] -> {mer: count<<1|hq} by the (H, L) rule"""
    mx = (1 << bits) - 1
    H, L = {}, {}
    for m, q in instances:
        d = H if q else L
        d[m] = d.get(m, 0) + 1
    res = {}
    for m in set(H) | set(L):
        if m in H:
            res[m] = (min(H[m], mx) << 1) | 1
        else:
            res[m] = min(L[m], mx) << 1
    return res


def dict_model_np(mers, hq, bits):
    """numpy form of dict_model -> (sorted mers, vals)"""
    mx = (1 << bits) - 1
    mers = np.asarray(mers, dtype=np.uint64)
    hq = np.asarray(hq, dtype=np.uint8)
    u, inv = np.unique(mers, return_inverse=True)
    H = np.bincount(inv, weights=(hq != 0), minlength=u.size).astype(np.uint64)
    L = np.bincount(inv, weights=(hq == 0), minlength=u.size).astype(np.uint64)
    c = np.where(H > 0, np.minimum(H, mx), np.minimum(L, mx)).astype(np.uint64)
    return u, (c << np.uint64(1)) | (H > 0).astype(np.uint64)


def split_fastq(data):
    """strict 4-line FASTQ -> [(seq, qual)]"""
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    assert len(lines) % 4 == 0
    return [(lines[i + 1], lines[i + 3]) for i in range(0, len(lines), 4)]


def sorted_pairs(mers, vals):
    mers = np.asarray(mers, dtype=np.uint64)
    vals = np.asarray(vals, dtype=np.uint64)
    o = np.argsort(mers, kind="stable")
    return mers[o], vals[o]


def random_fastq(rng, n_reads, max_len=160, genome=None, k=24):
    """strict 4-line FASTQ with N's, lowercase, IUPAC, quality dips, short and empty-ish reads,
    reads drawn from a small genome so that k-mers repeat"""
    if genome is None:
        genome = rng.integers(0, 4, size=5000)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    recs = []
    for i in range(n_reads):
        L = int(rng.integers(1, max_len + 1))
        if rng.random() < 0.05:
            L = int(rng.integers(1, k + 2))  # around k
        st = int(rng.integers(0, genome.size - L))
        codes = genome[st:st + L].copy()
        if rng.random() < 0.5:
            codes = 3 - codes[::-1]
        seq = bases[codes].copy()
        errs = rng.random(L) < 0.02
        seq[errs] = bases[rng.integers(0, 4, size=int(errs.sum()))]
        r = rng.random()
        if r < 0.3:
            seq[rng.random(L) < 0.02] = ord("N")
        elif r < 0.35:
            seq[rng.random(L) < 0.05] = rng.choice(np.frombuffer(b"RYKMSWnx.-*", dtype=np.uint8))
        if rng.random() < 0.1:
            low = rng.random(L) < 0.5
            seq[low] |= 0x20  # lowercase
        qual = np.full(L, ord("I"), dtype=np.uint8)
        if rng.random() < 0.6:
            dips = rng.random(L) < 0.03
            qual[dips] = rng.integers(33, 75, size=int(dips.sum()))
        if rng.random() < 0.2:
            T = int(rng.integers(0, L + 1))
            qual[L - T:] = ord("#")
        if rng.random() < 0.02:
            qual[rng.random(L) < 0.1] = 0xC9  # negative as signed char
        recs.append(b"@read%d some comment\n" % i + seq.tobytes() + b"\n+\n" + qual.tobytes() + b"\n")
    return b"".join(recs)
