"""Hash-sharded multi-GPU build: one process per GPU, k-mers
hash-partitioned by owner shard and exchanged with ONE variable-size all-to-all
(torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU tests), then inserted into the local shard. The reference has no counterpart
(single process, pthreads: src/create_database.cc:122).

torch is plumbing here: device buffers, the collective and streams. Extraction,
bucketing and insertion are the library's kernels (include/quorum_hip.h)."""
import ctypes as C
import math
import os

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


# Largest single message of an exchange, in items. Measured on this image (tools/rccl_self_check.py, RCCL 2.26.6, one
# rank sending to itself): messages up to 1 GiB arrive whole, a 1.6 GB one arrives with its second half missing and no
# error. Every exchange is therefore cut into rounds whose messages stay at or below 512 MiB.
MAX_MESSAGE_ITEMS = 1 << 26


def _max_message_items():
    return max(1, int(os.environ.get("QH_EXCHANGE_MAX_ITEMS", MAX_MESSAGE_ITEMS)))


def gather_counts(counts, group=None, device=None):
    """Every rank's per-destination counts -> the world x world matrix m[src][dst] as python lists (one small
    all-gather): each rank learns what it will receive AND how many rounds the largest message of the job needs."""
    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    sc = torch.tensor(list(counts), dtype=torch.int64, device=dev)
    rows = [torch.empty(world, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(rows, sc, group=group)
    return torch.stack(rows).tolist()


def _rounds_of(matrix, limit):
    return max(1, -(-max(max(r) for r in matrix) // limit))


_LIST_ALL_TO_ALL_OK = True


def _run_exchange(send_rows, counts, recv, recv_counts, rounds, limit, group, async_op=False):
    """send_rows[s]: 1-D tensor whose first counts[s] items go to rank s; recv: 1-D tensor that receives
    sum(recv_counts) items, source after source. Round t carries items [t*limit, (t+1)*limit) of every message.
    With RCCL the parts are sent and received where they lie (list form); otherwise each round is packed.
    Returns the list of pending work handles (empty unless async_op and the list form was taken)."""
    global _LIST_ALL_TO_ALL_OK
    world = len(counts)
    offs = [0]
    for c in recv_counts:
        offs.append(offs[-1] + c)
    nccl = recv.device.type == "cuda" and dist.get_backend(group) == "nccl"
    works = []
    for t in range(rounds):
        lo = [min(c, t * limit) for c in counts]
        hi = [min(c, (t + 1) * limit) for c in counts]
        rlo = [min(c, t * limit) for c in recv_counts]
        rhi = [min(c, (t + 1) * limit) for c in recv_counts]
        ins = [send_rows[s][lo[s]:hi[s]] for s in range(world)]
        outs = [recv[offs[s] + rlo[s]:offs[s] + rhi[s]] for s in range(world)]
        if nccl and _LIST_ALL_TO_ALL_OK:
            try:
                w = dist.all_to_all(outs, ins, group=group, async_op=async_op)
                if async_op:
                    works.append(w)
                continue
            except (RuntimeError, NotImplementedError):
                _LIST_ALL_TO_ALL_OK = False   # every rank fails alike (same torch build): fall back together
        packed = torch.cat(ins)
        got = torch.empty(sum(rhi[s] - rlo[s] for s in range(world)), dtype=recv.dtype, device=recv.device)
        dist.all_to_all_single(got, packed, output_split_sizes=[rhi[s] - rlo[s] for s in range(world)],
                               input_split_sizes=[hi[s] - lo[s] for s in range(world)], group=group)
        for o, g in zip(outs, got.split([rhi[s] - rlo[s] for s in range(world)])):
            o.copy_(g)
    return works


def exchange_items(send, send_counts, group=None):
    """One all-to-all(v) of packed items (mer<<1|hq as int64).

    send: 1-D int64 tensor, the items for rank 0, then rank 1, ... back to back;
    send_counts: python list of per-destination counts. Returns (recv, recv_counts).
    Works on CUDA tensors (RCCL) and CPU tensors (gloo)."""
    dev = send.device
    if dev.type == "cuda" and dist.get_backend(group) == "gloo":
        # rehearsal mode (several ranks sharing one GPU, no RCCL): stage through the host
        recv, recv_counts = exchange_items(send.cpu(), send_counts, group)
        return recv.to(dev), recv_counts
    rank = dist.get_rank(group)
    m = gather_counts(send_counts, group, dev)
    recv_counts = [row[rank] for row in m]
    recv = torch.empty(int(sum(recv_counts)), dtype=torch.int64, device=dev)
    limit = _max_message_items()
    _run_exchange(list(send.split(list(send_counts))), list(send_counts), recv, recv_counts, _rounds_of(m, limit), limit, group)
    return recv, recv_counts


def exchange_buckets(buckets, counts, group=None):
    """The same exchange straight from the per-destination bucket rows (2-D int64 tensor, counts[s] items of row s
    are for rank s): with RCCL the rows are sent where they lie, which saves packing 8 bytes per item into one send
    buffer."""
    dev = buckets.device
    if dev.type == "cuda" and dist.get_backend(group) == "gloo":
        recv, recv_counts = exchange_buckets(buckets.cpu(), counts, group)
        return recv.to(dev), recv_counts
    rank = dist.get_rank(group)
    m = gather_counts(counts, group, dev)
    recv_counts = [row[rank] for row in m]
    recv = torch.empty(int(sum(recv_counts)), dtype=torch.int64, device=dev)
    limit = _max_message_items()
    _run_exchange([buckets[s] for s in range(len(counts))], list(counts), recv, recv_counts, _rounds_of(m, limit), limit, group)
    return recv, recv_counts


class ShardedBuilder:
    """Per-rank driver of the hash-sharded build: every piece of reads goes through three stages,

        extract + bucket by owner shard  ->  all-to-all exchange  ->  insert what this shard owns,

    and the stages of consecutive pieces overlap: while piece i travels (RCCL's own stream), piece i+1 is
    extracted (the extraction handle's stream) and piece i-1 is inserted (the shard handle's stream). Counts travel
    ahead of the items in one small all-to-all. A piece whose buckets overflow (a skewed chunk) is extracted again
    with twice the bucket capacity instead of failing.

    All ranks must feed the same number of pieces (a rank that has run out of reads passes None); finish() /
    flush() drain the pipeline."""

    def __init__(self, k, bits, total_size, min_qual, rank, world, device, reprobe_limit=126, group=None,
                 cap_factor=1.25, force_pipeline=False, own_stream=False):
        from .mer_database import HashWithQuality
        self.cap_factor = cap_factor     # bucket capacity = mean share x this (tests lower it to force the retry)
        assert world & (world - 1) == 0, "number of shards must be a power of two"
        self.k, self.rank, self.world, self.group = k, rank, world, group
        self.shard_bits = int(math.log2(world))
        self.device = torch.device("cuda", device)
        self.h = HashWithQuality(total_size, 2 * k, bits, min_qual=min_qual, device=device,
                                 reprobe_limit=reprobe_limit, shard_bits=self.shard_bits, shard_id=rank)
        self.hx = None           # extraction handle (its table is never used: smallest size)
        self._ins_stream = None
        # force_pipeline: a single rank takes the extract / exchange / insert route too (the collective then sends to
        # itself): the way to run the RCCL calls of this file on a one-GPU machine
        self.pipelined = world > 1 or force_pipeline
        if self.pipelined:
            self.hx = HashWithQuality(1 << 13, 2 * k, bits, min_qual=min_qual, device=device,
                                      reprobe_limit=reprobe_limit, shard_bits=self.shard_bits, shard_id=rank)
            self._ins_stream = torch.cuda.Stream(self.device)
            self.h.set_stream(self._ins_stream.cuda_stream)
        elif not own_stream:
            # the chunks are torch tensors: on torch's stream their text is ordered behind whatever produced it.
            # own_stream: the caller has synchronised (bench.py draws its reads before the timed region); the handle
            # then keeps its own stream and may prepare chunk i+1 on a second one while chunk i is still being split
            self.h.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self._buckets = [None, None]     # send side, two pieces in flight
        self._counts = torch.zeros(world, dtype=torch.int64, device=self.device)
        self._recv = [None, None]        # receive side
        self._sent_done = [None, None]   # event: the exchange that read _buckets[j] has completed
        self._inflight = None            # (piece index, work handle or None, recv tensor, recv count)
        self._npieces = 0
        self.retries = 0                 # bucket overflows that were resolved by a larger capacity
        self.rounds_max = 0              # most rounds any exchange needed (messages are capped, see MAX_MESSAGE_ITEMS)

    # ---- stage 1
    def _extract(self, chunk, max_items, slot):
        """-> per-destination counts; the items lie in self._buckets[slot]"""
        from ._lib import lib, check, QhError, QH_EBADARG, last_error
        if chunk is None or chunk.numel() == 0:
            return [0] * self.world
        nbytes = chunk.numel()
        if max_items is None:
            max_items = nbytes // 2
        cap = int(max_items / self.world * self.cap_factor) + 4096
        while True:
            b = self._buckets[slot]
            if b is None or b.shape[1] < cap:
                b = self._buckets[slot] = torch.empty((self.world, cap), dtype=torch.int64, device=self.device)
                torch.cuda.current_stream(self.device).synchronize()
            st = lib.qh_build_extract_partitioned_device(self.hx._h, C.c_void_p(chunk.data_ptr()), nbytes,
                                                         C.c_void_p(b.data_ptr()), b.shape[1],
                                                         C.c_void_p(self._counts.data_ptr()))
            if st == QH_EBADARG and "cap_per_shard too small" in last_error():
                cap = 2 * b.shape[1]        # a skewed piece: again with room to spare
                self.retries += 1
                continue
            check(st)
            return self._counts.tolist()

    # ---- stage 2
    def _start_exchange(self, slot, counts):
        """counts ahead (every rank learns the whole matrix), then the items in as many rounds as the job's largest
        message needs; returns (list of work handles, recv tensor, number received)"""
        world, dev = self.world, self.device
        staged = dist.get_backend(self.group) != "nccl"     # rehearsal (several ranks on one GPU over gloo): through the host
        m = gather_counts(counts, self.group, torch.device("cpu") if staged else dev)
        recv_counts = [row[self.rank] for row in m]
        limit = _max_message_items()
        rounds = _rounds_of(m, limit)
        self.rounds_max = max(self.rounds_max, rounds)
        n = int(sum(recv_counts))
        r = self._recv[slot]
        if r is None or r.numel() < n:
            r = self._recv[slot] = torch.empty(max(n + n // 8, 1), dtype=torch.int64, device=dev)
        b = self._buckets[slot]
        if b is None:                  # this rank had nothing to extract yet
            b = self._buckets[slot] = torch.empty((world, 1), dtype=torch.int64, device=dev)
        if staged:   # gloo: host tensors
            got = torch.empty(n, dtype=torch.int64)
            _run_exchange([b[s, :counts[s]].cpu() for s in range(world)], counts, got, recv_counts, rounds, limit, self.group)
            r[:n].copy_(got)
            torch.cuda.current_stream(dev).synchronize()
            return [], r, n
        works = _run_exchange([b[s] for s in range(world)], counts, r[:n], recv_counts, rounds, limit, self.group, async_op=True)
        # the packed fallback works on the current stream: the shard handle's stream must follow that too
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self._ins_stream.wait_event(ev)
        return works, r, n

    # ---- stage 3
    def _insert(self, slot, works, recv, n):
        if works is not None:
            with torch.cuda.stream(self._ins_stream):
                for w in works:
                    w.wait()                      # the shard handle's stream waits for the items; the host does not
                ev = torch.cuda.Event()
                ev.record(self._ins_stream)
            self._sent_done[slot] = ev
        if n:
            self.h.add_hashed_device(recv.data_ptr(), n)   # waits (host) for the previous piece's insert, then enqueues
        else:
            self.h.sync()                                  # same guarantee for the buffers when nothing arrived

    def add_fastq_device(self, chunk, max_items=None):
        """chunk: 1-D uint8 CUDA tensor holding strict 4-line FASTQ records (None: this rank has no more reads but
        other ranks may). The chunk must stay alive until the next call returns."""
        if not self.pipelined:
            if chunk is not None and chunk.numel():
                self.h.add_fastq_device(chunk.data_ptr(), chunk.numel())
            return
        i = self._npieces
        slot = i & 1
        if self._sent_done[slot] is not None:     # the exchange that read this send buffer two pieces ago
            self._sent_done[slot].synchronize()
            self._sent_done[slot] = None
        counts = self._extract(chunk, max_items, slot)       # overlaps piece i-1 on the wire and piece i-2 being inserted
        prev = self._inflight
        self._inflight = None
        if prev is not None:
            self._insert(*prev)                              # piece i-1: its items have been requested a stage ago
        work, recv, n = self._start_exchange(slot, counts)
        self._inflight = (slot, work, recv, n)
        self._npieces += 1

    def flush(self):
        """drains the pipeline: every piece fed so far is in the table (or queued on the shard handle's stream)"""
        prev, self._inflight = self._inflight, None
        if prev is not None:
            self._insert(*prev)
        if not self.pipelined:
            self.h.flush()   # the pass over the table starts; whoever collects next waits for it

    def reset(self):
        self.flush()
        self.h.reset()

    def finish(self):
        self.flush()
        return self.h.done()
