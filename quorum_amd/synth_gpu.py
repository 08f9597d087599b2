"""Synthetic Illumina-like reads generated ON THE GPU with torch (same shape as quorum_amd/synth.py, which makes them
with numpy on the host): inputs of bench.py's multi-GPU workload (BASELINE configs[2]: 50 M reads, too many to draw on
the host inside a benchmark run) and of the full-size GPU tests. Input generation only -- nothing here is on the
measured path.

Genome: uniform i.i.d. ACGT. Reads: uniform start, uniform strand, substitutions at rate `err` to one of the three other
bases, 'N' at rate `n_rate`, Phred+33 qualities 'I' (40) with substituted bases at U[2,15]; optional low-quality tails.
Every record is strict 4-line FASTQ of the same byte length: '@r<9 digits>\\n<L bases>\\n+\\n<L quals>\\n' (2L + 16 bytes)."""
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


