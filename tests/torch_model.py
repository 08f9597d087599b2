"""Full-size test infrastructure on the GPU (torch tensors only, nothing of libquorum_hip): a third, independent
model of the path -- every window of k valid bases -> (min(mer, revcomp), hq) (src/create_database.cc:64-92) -> (H, L)
dictionary rule (src/mer_database.hpp:102-111) -- written with whole-tensor int64 arithmetic, so that the
BASELINE configurations can be checked at their real per-GPU sizes in seconds. The read generator is
quorum_amd/synth_gpu.py. Checker code: never on the measured path."""
import numpy as np
import torch

from quorum_amd.synth_gpu import make_genome, make_reads, fastq_text  # noqa: F401  (re-exported for the tests)


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
