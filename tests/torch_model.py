"""Full-size test infrastructure on the GPU (torch tensors only, nothing of libquorum_hip): a read generator with
the shape of quorum_amd/synth.py and a third, independent model of the path -- every window of k valid bases
-> (min(mer, revcomp), hq) (src/create_database.cc:64-92) -> (H, L) dictionary rule (src/mer_database.hpp:102-111,
SURVEY fact 2) -- written with whole-tensor int64 arithmetic, so that the BASELINE configurations can be checked
at their real per-GPU sizes in seconds. Checker code: never on the measured path."""
import numpy as np
import torch


def make_genome(n_bases, device, seed=42):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.randint(0, 4, (n_bases,), generator=g, device=device, dtype=torch.uint8)


def make_reads(genome, n, L, seed, err=0.01, n_rate=0.0005, tails=False):
    """-> (codes uint8 [n, L], qual uint8 [n, L], isN bool [n, L])"""
    dev = genome.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    starts = torch.randint(0, genome.numel() - L + 1, (n,), generator=g, device=dev)
    ar = torch.arange(L, device=dev)
    codes = genome[starts[:, None] + ar[None, :]]
    rc = torch.rand(n, generator=g, device=dev) < 0.5
    codes = torch.where(rc[:, None], 3 - codes.flip(1), codes)
    errs = torch.rand((n, L), generator=g, device=dev) < err
    sub = torch.randint(1, 4, (n, L), generator=g, device=dev, dtype=torch.uint8)
    codes = torch.where(errs, (codes + sub) & 3, codes)
    qual = torch.full((n, L), ord("I"), dtype=torch.uint8, device=dev)
    if tails:
        T = torch.randint(0, 41, (n,), generator=g, device=dev)
        j = ar[None, :] - (L - T[:, None])
        q = 40.0 - (j + 1).to(torch.float32) * 38.0 / T.clamp(min=1)[:, None].to(torch.float32)
        qt = (q.round().clamp(2, 40) + 33).to(torch.uint8)
        qual = torch.where(j >= 0, qt, qual)
    lowq = torch.randint(2, 16, (n, L), generator=g, device=dev, dtype=torch.uint8) + 33
    qual = torch.where(errs, lowq, qual)
    isN = torch.rand((n, L), generator=g, device=dev) < n_rate
    return codes, qual, isN


def fastq_text(codes, qual, isN, first_id=0):
    """strict 4-line records '@r<9 digits>\\n<bases>\\n+\\n<quals>\\n' as one uint8 tensor"""
    n, L = codes.shape
    dev = codes.device
    rec = torch.empty((n, 2 * L + 16), dtype=torch.uint8, device=dev)
    ids = first_id + torch.arange(n, device=dev, dtype=torch.int64)
    pow10 = 10 ** torch.arange(8, -1, -1, device=dev, dtype=torch.int64)
    rec[:, 0] = ord("@")
    rec[:, 1] = ord("r")
    rec[:, 2:11] = ((ids[:, None] // pow10[None, :]) % 10 + 48).to(torch.uint8)
    rec[:, 11] = 10
    letters = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    bases = letters[codes.long()]
    rec[:, 12:12 + L] = torch.where(isN, torch.full_like(bases, ord("N")), bases)
    rec[:, 12 + L] = 10
    rec[:, 13 + L] = ord("+")
    rec[:, 14 + L] = 10
    rec[:, 15 + L:15 + 2 * L] = qual
    rec[:, 15 + 2 * L] = 10
    return rec.reshape(-1)


def model_items(codes, qual, isN, k, thresh):
    """every window of k valid bases -> (canonical mer as int64, hq bool)"""
    n, L = codes.shape
    W = L - k + 1
    c = codes.to(torch.int64)
    m = torch.zeros((n, W), dtype=torch.int64, device=codes.device)
    r = torch.zeros((n, W), dtype=torch.int64, device=codes.device)
    for j in range(k):
        m = (m << 2) | c[:, j:j + W]                           # first base of the window most significant
        r = (r << 2) | (3 - c[:, k - 1 - j:k - 1 - j + W])     # revcomp: complement of the last base most significant
    canon = torch.minimum(m, r)
    del m, r, c

    def window_sum(flag):
        cs = torch.cumsum(flag.to(torch.int32), dim=1)
        z = torch.zeros((n, 1), dtype=torch.int32, device=codes.device)
        cs = torch.cat([z, cs], dim=1)
        return cs[:, k:] - cs[:, :W]

    valid = window_sum(isN) == 0
    t = thresh - 256 if thresh >= 128 else thresh
    sq = qual.to(torch.int16)
    sq = torch.where(sq >= 128, sq - 256, sq)
    hq = window_sum(sq < t) == 0
    return canon[valid], hq[valid]


def model_table(canon, hq, bits):
    """(H, L) rule -> (sorted mers int64, vals int64 = count << 1 | hq)"""
    mx = (1 << bits) - 1
    uH, cH = torch.unique(canon[hq], return_counts=True)
    uL, cL = torch.unique(canon[~hq], return_counts=True)
    only_l = ~torch.isin(uL, uH, assume_unique=True)
    mers = torch.cat([uH, uL[only_l]])
    vals = torch.cat([(cH.clamp(max=mx) << 1) | 1, cL[only_l].clamp(max=mx) << 1])
    o = torch.argsort(mers)
    return mers[o], vals[o]


def dumps_to_sorted(dumps, device):
    """[(mers uint64 ndarray, vals uint64 ndarray)] -> sorted int64 tensors on the device"""
    m = torch.from_numpy(np.concatenate([d[0] for d in dumps]).view(np.int64)).to(device)
    v = torch.from_numpy(np.concatenate([d[1] for d in dumps]).view(np.int64)).to(device)
    o = torch.argsort(m)
    return m[o], v[o]
