"""Hash-sharded multi-GPU build: one process per GPU, k-mers
hash-partitioned by owner shard and exchanged with ONE variable-size all-to-all
(torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU tests), then inserted into the local shard. The reference has no counterpart
(single process, pthreads: src/create_database.cc:122).

torch is plumbing here: device buffers, the collective and streams. Extraction,
bucketing and insertion are the library's kernels (include/quorum_hip.h)."""
import ctypes as C
import math

import numpy as np
import torch
import torch.distributed as dist

MIX_C1 = np.uint64(0x9E3779B1)
MIX_C2 = np.uint64(0x85EBCA77)
MIX_C3 = np.uint64(0xC2B2AE3D)


def _compress_even(x):
    x = x & np.uint64(0x5555555555555555)
    for sh, m in ((1, 0x3333333333333333), (2, 0x0F0F0F0F0F0F0F0F), (4, 0x00FF00FF00FF00FF),
                  (8, 0x0000FFFF0000FFFF), (16, 0x00000000FFFFFFFF)):
        x = (x | (x >> np.uint64(sh))) & np.uint64(m)
    return x


def _spread_even(x):
    x = x & np.uint64(0xFFFFFFFF)
    for sh, m in ((16, 0x0000FFFF0000FFFF), (8, 0x00FF00FF00FF00FF), (4, 0x0F0F0F0F0F0F0F0F),
                  (2, 0x3333333333333333), (1, 0x5555555555555555)):
        x = (x | (x << np.uint64(sh))) & np.uint64(m)
    return x


def pkey_np(mers, k):
    """qh_pkey: 2-bit packed mers -> plane keys (low code bits | high code bits << k)"""
    mers = np.asarray(mers, dtype=np.uint64)
    return _compress_even(mers) | (_compress_even(mers >> np.uint64(1)) << np.uint64(k))


def key_from_pkey_np(pk, k):
    pk = np.asarray(pk, dtype=np.uint64)
    km = np.uint64((1 << k) - 1)
    return _spread_even(pk & km) | (_spread_even((pk >> np.uint64(k)) & km) << np.uint64(1))


MIX_Z = np.uint64(5)
_M32 = np.uint64(0xFFFFFFFF)


def _feistel3(L, R, n):
    nn, sh = np.uint64(n), np.uint64(32 - n)
    R = R ^ (((L * MIX_C1) & _M32) >> sh)
    L = L ^ (((R * MIX_C2) & _M32) >> sh)
    R = R ^ (((L * MIX_C3) & _M32) >> sh)
    return R | (L << nn)


def _unfeistel3(x, n):
    nm, nn, sh = np.uint64((1 << n) - 1), np.uint64(n), np.uint64(32 - n)
    R, L = x & nm, (x >> nn) & nm
    R = R ^ (((L * MIX_C3) & _M32) >> sh)
    L = L ^ (((R * MIX_C2) & _M32) >> sh)
    R = R ^ (((L * MIX_C1) & _M32) >> sh)
    return L, R


def _rc_planes(L, R, n):
    """planes of the reverse complement of an n-base sequence (base i at bit i)"""
    rL, rR = np.zeros_like(L), np.zeros_like(R)
    one = np.uint64(1)
    for i in range(n):
        rL |= ((~(L >> np.uint64(i))) & one) << np.uint64(n - 1 - i)
        rR |= ((~(R >> np.uint64(i))) & one) << np.uint64(n - 1 - i)
    return rL, rR


def _packed(L, R):
    return _spread_even(L) | (_spread_even(R) << np.uint64(1))


def mix_np(x, k):
    """qh_mix (quorum_amd/csrc/qh_spec.h, "feistel3-core/2") on a uint64 array of plane keys:
    h = [flip | 4 end-base bits | Feistel(canonical core) ^ Z]; the low bits (shard, window, home slot)
    depend only on the canonical core = bases 1..k-2 in their smaller orientation"""
    x = np.asarray(x, dtype=np.uint64)
    km, kk = np.uint64((1 << k) - 1), np.uint64(k)
    L, R = x & km, (x >> kk) & km
    if k < 4:
        return _feistel3(L, R, k)
    kc = k - 2
    cm, one = np.uint64((1 << kc) - 1), np.uint64(1)
    cL, cR = (L >> one) & cm, (R >> one) & cm
    dL, dR = _rc_planes(cL, cR, kc)
    flip = _packed(cL, cR) > _packed(dL, dR)
    cL, cR = np.where(flip, dL, cL), np.where(flip, dR, cR)
    top = np.uint64(k - 1)
    e = (L & one) | ((R & one) << one) | ((L >> top) << np.uint64(2)) | ((R >> top) << np.uint64(3))
    return (_feistel3(cL, cR, kc) ^ MIX_Z) | (e << np.uint64(2 * kc)) | (flip.astype(np.uint64) << np.uint64(2 * kc + 4))


def unmix_np(h, k):
    h = np.asarray(h, dtype=np.uint64)
    if k < 4:
        L, R = _unfeistel3(h, k)
        return L | (R << np.uint64(k))
    kc = k - 2
    one = np.uint64(1)
    cL, cR = _unfeistel3((h & np.uint64((1 << (2 * kc)) - 1)) ^ MIX_Z, kc)
    e = (h >> np.uint64(2 * kc)) & np.uint64(15)
    flip = ((h >> np.uint64(2 * kc + 4)) & one) != 0
    dL, dR = _rc_planes(cL, cR, kc)
    cL, cR = np.where(flip, dL, cL), np.where(flip, dR, cR)
    top = np.uint64(k - 1)
    L = (cL << one) | (e & one) | (((e >> np.uint64(2)) & one) << top)
    R = (cR << one) | ((e >> one) & one) | (((e >> np.uint64(3)) & one) << top)
    return L | (R << np.uint64(k))


def hash_np(mers, k):
    """qh_hash_key: the position hash of 2-bit packed canonical mers"""
    return mix_np(pkey_np(mers, k), k)


def unhash_np(h, k):
    return key_from_pkey_np(unmix_np(h, k), k)


def owner_np(mers, k, shard_bits):
    """owner shard of 2-bit packed canonical mers = low shard_bits of their hash"""
    return (hash_np(mers, k) & np.uint64((1 << shard_bits) - 1)).astype(np.int64)


def exchange_items(send, send_counts, group=None):
    """One all-to-all(v) of packed items (mer<<1|hq as int64).

    send: 1-D int64 tensor, the items for rank 0, then rank 1, ... back to back;
    send_counts: python list of per-destination counts. Returns (recv, recv_counts).
    Works on CUDA tensors (RCCL) and CPU tensors (gloo)."""
    world = dist.get_world_size(group)
    dev = send.device
    if dev.type == "cuda" and dist.get_backend(group) == "gloo":
        # rehearsal mode (several ranks sharing one GPU, no RCCL): stage through the host
        recv, recv_counts = exchange_items(send.cpu(), send_counts, group)
        return recv.to(dev), recv_counts
    sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
    rc = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(rc, sc, group=group)
    recv_counts = rc.tolist()
    recv = torch.empty(int(sum(recv_counts)), dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=list(send_counts), group=group)
    return recv, recv_counts


_LIST_ALL_TO_ALL_OK = True


def exchange_buckets(buckets, counts, group=None):
    """The same exchange straight from the per-destination bucket rows (2-D int64 CUDA tensor,
    counts[s] items of row s are for rank s): with RCCL the rows are sent where they lie
    (dist.all_to_all over views), which saves packing 8 bytes per item into one send buffer; any
    other backend, or a torch build that refuses the list form, takes exchange_items."""
    global _LIST_ALL_TO_ALL_OK
    world = dist.get_world_size(group)
    if _LIST_ALL_TO_ALL_OK and buckets.device.type == "cuda" and dist.get_backend(group) == "nccl":
        try:
            sc = torch.tensor(counts, dtype=torch.int64, device=buckets.device)
            rc = torch.empty(world, dtype=torch.int64, device=buckets.device)
            dist.all_to_all_single(rc, sc, group=group)
            recv_counts = rc.tolist()
            recv = torch.empty(int(sum(recv_counts)), dtype=torch.int64, device=buckets.device)
            dist.all_to_all(list(recv.split(recv_counts)), [buckets[s, :counts[s]] for s in range(world)], group=group)
            return recv, recv_counts
        except (RuntimeError, NotImplementedError):
            _LIST_ALL_TO_ALL_OK = False  # every rank fails alike (same torch build): fall back together
    send = torch.cat([buckets[s, :counts[s]] for s in range(world)])
    return exchange_items(send, counts, group)


class ShardedBuilder:
    """Per-rank driver: extract+bucket the local chunk, exchange, insert what this shard owns."""

    def __init__(self, k, bits, total_size, min_qual, rank, world, device, reprobe_limit=126, group=None):
        from .mer_database import HashWithQuality
        assert world & (world - 1) == 0, "number of shards must be a power of two"
        self.k, self.rank, self.world, self.group = k, rank, world, group
        self.shard_bits = int(math.log2(world))
        self.device = torch.device("cuda", device)
        self.h = HashWithQuality(total_size, 2 * k, bits, min_qual=min_qual, device=device,
                                 reprobe_limit=reprobe_limit, shard_bits=self.shard_bits, shard_id=rank)
        self.h.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self._buckets = None
        self._counts = torch.zeros(world, dtype=torch.int64, device=self.device)

    def _ensure(self, cap):
        if self._buckets is None or self._buckets.shape[1] < cap:
            self._buckets = torch.empty((self.world, cap), dtype=torch.int64, device=self.device)

    def add_fastq_device(self, chunk, max_items=None):
        """chunk: 1-D uint8 CUDA tensor holding strict 4-line FASTQ records"""
        from ._lib import lib, check
        nbytes = chunk.numel()
        if max_items is None:
            max_items = nbytes // 2
        if self.world == 1:
            self.h.add_fastq_device(chunk.data_ptr(), nbytes)
            return
        cap = int(max_items / self.world * 1.25) + 4096
        self._ensure(cap)
        check(lib.qh_build_extract_partitioned_device(self.h._h, C.c_void_p(chunk.data_ptr()), nbytes,
                                                      C.c_void_p(self._buckets.data_ptr()), self._buckets.shape[1],
                                                      C.c_void_p(self._counts.data_ptr())))
        counts = self._counts.tolist()
        recv, _ = exchange_buckets(self._buckets, counts, self.group)
        if recv.numel():
            self.h.add_hashed_device(recv.data_ptr(), recv.numel())
            self.h.sync()  # recv must stay alive until the insert kernel has consumed it

    def reset(self):
        self.h.reset()

    def finish(self):
        return self.h.done()
