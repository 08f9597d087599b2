"""BASELINE configs[2] and configs[3] at their real per-GPU sizes on the one GPU of the box, through the C ABI.

The eight ranks of the sharded build are eight shard handles in one process; the exchange between them is a
pointer hand-over (the RCCL leg is covered over gloo by tests/test_dist_cpu.py). What a rank does at these
configurations is done in full: its share of the reads in several chunks through the owner-bucketing extraction,
shards of the configuration's size (2^28 slots at configs[2], 2^32 at configs[3]) updated by several chunks (the
window pass over a table that already holds entries). Checked bit-exactly, at full size, against the independent
whole-tensor model of tests/torch_model.py (pinned to the oracle by tests/test_torch_model.py), plus the
size-independent properties: instance count, order independence, doubling."""
import ctypes as C
import time

import numpy as np
import pytest
import torch

import quorum_amd as qa
from quorum_amd import _lib
import torch_model as tm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def extract_and_insert(hs, chunk, cap, shard_bits):
    """what one step of the sharded build does on a rank, the all-to-all being a hand-over of bucket pointers"""
    n = 1 << shard_bits
    buckets = torch.empty((n, cap), dtype=torch.int64, device=DEV)
    counts = torch.zeros(n, dtype=torch.int64, device=DEV)
    torch.cuda.synchronize()   # the library works on its own stream: torch's writes of chunk / counts must have landed
    _lib.check(_lib.lib.qh_build_extract_partitioned_device(hs[0]._h, C.c_void_p(chunk.data_ptr()), chunk.numel(),
                                                            C.c_void_p(buckets.data_ptr()), cap, C.c_void_p(counts.data_ptr())))
    cnt = counts.tolist()
    for s in range(n):
        hs[s].add_hashed_device(buckets[s].data_ptr(), cnt[s])
    for h in hs:
        h.sync()
    return cnt


def check_tables(hs, model_m, model_v):
    dumps = [h.dump() for h in hs]
    m, v = tm.dumps_to_sorted(dumps, DEV)
    assert m.numel() == model_m.numel(), (m.numel(), model_m.numel())
    assert bool((m == model_m).all()) and bool((v == model_v).all())
    return sum(d[0].size for d in dumps)


def test_config2_per_gpu_share_k24_sharded_multichunk():
    """configs[2]: 100 Mbp genome, 50 M x 100 bp, k=24, 8 shards of 2^28 slots (-s 2G in all): one rank's 6.25 M
    reads in two chunks, every shard updated twice."""
    k, bits, thresh, L, shard_bits = 24, 7, 38, 100, 3
    genome = tm.make_genome(100_000_000, DEV)
    hs = [qa.HashWithQuality(1 << 31, 2 * k, bits, min_qual=thresh, shard_bits=shard_bits, shard_id=s) for s in range(8)]
    for h in hs:
        assert h.stats()["slots"] == 1 << 28
        h.set_mode("partition")      # the second, smaller chunk too: the window pass over a table that holds entries
        h.enable_timing(True)
    canon, hq, inst = [], [], 0
    t_chunks = []
    for ci, n_reads in enumerate((3_750_000, 2_500_000)):
        codes, qual, isN = tm.make_reads(genome, n_reads, L, seed=100 + ci)
        chunk = tm.fastq_text(codes, qual, isN, first_id=ci * 10_000_000)
        c, q = tm.model_items(codes, qual, isN, k, thresh)
        canon.append(c)
        hq.append(q)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cnt = extract_and_insert(hs, chunk, int(n_reads * (L - k + 1) / 8 * 1.25) + 4096, shard_bits)
        t_chunks.append(time.perf_counter() - t0)
        assert sum(cnt) == c.numel()                     # every window of k valid bases is one instance
        assert max(cnt) < 1.05 * min(cnt)                # the owner hash spreads them evenly
        inst += sum(cnt)
        del codes, qual, isN, chunk
    st = [h.stats() for h in hs]
    assert sum(s["instances"] for s in st) == inst and all(s["foreign"] == 0 and s["grows"] == 0 for s in st)
    mm, mv = tm.model_table(torch.cat(canon), torch.cat(hq), bits)
    del canon, hq
    n = check_tables(hs, mm, mv)
    assert n == sum(s["distinct"] for s in st)
    win_ms = [hs[0].kernel_time(3)[0]]
    print("configs[2] share: %d instances, %d distinct in 8 x 2^28 slots; chunk wall %s s; shard 0 window pass total %.2f ms"
          % (inst, n, ["%.3f" % t for t in t_chunks], win_ms[0]))
    for h in hs:
        h.close()


def test_config3_k31_sharded_2pow32_shard_multichunk():
    """configs[3]: k=31, 150 bp reads, 8 shards, 2^32 slots per GPU: shard 0 at its real size (34 GB), fed by two
    chunks of 3 M reads bucketed by owner; the seven other shards (2^28 slots each here) take the rest. The
    order-independence and doubling properties are checked on the big shard, the content of all eight against
    the model."""
    k, bits, thresh, L, shard_bits = 31, 7, 38, 150, 3
    genome = tm.make_genome(200_000_000, DEV)
    hs = [qa.HashWithQuality((1 << 35) if s == 0 else (1 << 31), 2 * k, bits, min_qual=thresh, shard_bits=shard_bits,
                             shard_id=s) for s in range(8)]
    assert hs[0].stats()["slots"] == 1 << 32
    for h in hs:
        h.set_mode("partition")
        h.enable_timing(True)
    canon, hq, inst, kept = [], [], 0, []
    for ci, n_reads in enumerate((3_000_000, 3_000_000)):
        codes, qual, isN = tm.make_reads(genome, n_reads, L, seed=300 + ci, tails=True)
        chunk = tm.fastq_text(codes, qual, isN, first_id=ci * 10_000_000)
        c, q = tm.model_items(codes, qual, isN, k, thresh)
        canon.append(c)
        hq.append(q)
        cnt = extract_and_insert(hs, chunk, int(n_reads * (L - k + 1) / 8 * 1.25) + 4096, shard_bits)
        assert sum(cnt) == c.numel()
        inst += sum(cnt)
        kept.append(chunk)
        del codes, qual, isN
    st = [h.stats() for h in hs]
    assert st[0]["slots"] == 1 << 32 and st[0]["grows"] == 0
    assert sum(s["instances"] for s in st) == inst
    mm, mv = tm.model_table(torch.cat(canon), torch.cat(hq), bits)
    del canon, hq
    check_tables(hs, mm, mv)
    for h in hs[1:]:
        h.close()
    # shard 0 again, from the raw text (owner check inside the extraction), chunks in the other order: same content
    m0, v0 = hs[0].dump()
    o = np.argsort(m0)
    m0, v0 = m0[o], v0[o]
    hs[0].reset()
    for chunk in reversed(kept):
        hs[0].add_fastq_device(chunk.data_ptr(), chunk.numel())
        hs[0].sync()
    st0 = hs[0].stats()
    assert st0["instances"] + st0["foreign"] == inst and st0["instances"] == st[0]["instances"]
    m1, v1 = hs[0].dump()
    o = np.argsort(m1)
    assert (m1[o] == m0).all() and (v1[o] == v0).all()
    # everything once more: counts double (saturating at 127), quality bits stay
    for chunk in kept:
        hs[0].add_fastq_device(chunk.data_ptr(), chunk.numel())
    hs[0].sync()
    m2, v2 = hs[0].dump()
    o = np.argsort(m2)
    m2, v2 = m2[o], v2[o]
    assert (m2 == m0).all() and ((v2 & np.uint64(1)) == (v0 & np.uint64(1))).all()
    assert ((v2 >> np.uint64(1)) == np.minimum(2 * (v0 >> np.uint64(1)), np.uint64(127))).all()
    print("configs[3] share: %d instances over 8 shards; shard 0: %d distinct in 2^32 slots, window pass total %.2f ms"
          % (inst, m0.size, hs[0].kernel_time(3)[0]))
    hs[0].close()
