"""The N>1 path on CPU: world_size-2 (and 4) gloo processes run the same exchange code
the GPU ranks run over RCCL (quorum_amd.dist_build.exchange_items), with the oracle
standing in for the device kernels on either side of the collective. Checks that the
owner function partitions the k-mer space, that every item reaches exactly its owner,
and that the union of the shard tables equals the single-process table."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_items(rank, world, k):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pyoracle as po
    from helpers import random_fastq, split_fastq
    rng = np.random.default_rng(100 + rank)
    genome = np.random.default_rng(99).integers(0, 4, size=20000)  # shared genome: shards see the same k-mers
    data = random_fastq(rng, 800, k=k, genome=genome)
    mers, hqs = [], []
    for seq, qual in split_fastq(data):
        m, h = po.extract_read(seq, qual, k, 38)
        mers.append(m)
        hqs.append(h)
    return np.concatenate(mers), np.concatenate(hqs)


def _worker(rank, world, port, k, bits, outdir, max_items=0):
    if max_items:
        os.environ["QH_EXCHANGE_MAX_ITEMS"] = str(max_items)   # messages cut into rounds of this many items
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from quorum_amd.dist_build import exchange_items, owner_np
    import pyoracle as po
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        shard_bits = world.bit_length() - 1
        mers, hq = _rank_items(rank, world, k)
        owner = owner_np(mers, k, shard_bits)
        order = np.argsort(owner, kind="stable")
        items = ((mers << np.uint64(1)) | hq.astype(np.uint64))[order]
        counts = np.bincount(owner, minlength=world).tolist()
        send = torch.from_numpy(items.view(np.int64).copy())
        recv, recv_counts = exchange_items(send, counts)
        got = recv.numpy().view(np.uint64)
        assert got.size == sum(recv_counts)
        if max_items:   # the same through the bucket-row form, which must deliver the same items in the same order
            from quorum_amd.dist_build import exchange_buckets
            rows = torch.zeros((world, max(counts) + 3), dtype=torch.int64)
            for s_, part in enumerate(send.split(counts)):
                rows[s_, :counts[s_]] = part
            recv2, rc2 = exchange_buckets(rows, counts)
            assert rc2 == recv_counts and torch.equal(recv2, recv)
        gm, gq = got >> np.uint64(1), (got & np.uint64(1)).astype(np.uint8)
        assert (owner_np(gm, k, shard_bits) == rank).all()      # only what this shard owns arrives
        db = po.OracleDB(1 << 16, k, bits)
        db.add_mers(gm, gq)
        m, v = db.sorted_dump()
        np.savez(os.path.join(outdir, "shard%d.npz" % rank), m=m, v=v, n=np.array([got.size]), sent=np.array([items.size]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,max_items", [(2, 0), (4, 0), (2, 7001), (4, 1000)])
def test_sharded_exchange_gloo(world, max_items, tmp_path):
    """max_items: the cap on a single message (dist_build.MAX_MESSAGE_ITEMS), lowered so that the ~15-30 k items a pair
    exchanges here travel in several rounds, the last one ragged"""
    k, bits = 21, 7
    port = _free_port()
    mp.spawn(_worker, args=(world, port, k, bits, str(tmp_path), max_items), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    ref = po.OracleDB(1 << 16, k, bits)
    total = 0
    for r in range(world):
        m, q = _rank_items(r, world, k)
        ref.add_mers(m, q)
        total += m.size
    rm, rv = ref.sorted_dump()
    shards = [np.load(os.path.join(str(tmp_path), "shard%d.npz" % r)) for r in range(world)]
    assert sum(int(s["n"][0]) for s in shards) == total == sum(int(s["sent"][0]) for s in shards)
    m = np.concatenate([s["m"] for s in shards])
    v = np.concatenate([s["v"] for s in shards])
    assert np.unique(m).size == m.size                          # shards are disjoint
    o = np.argsort(m)
    assert (m[o] == rm).all() and (v[o] == rv).all()            # union == single-process table
    sizes = [s["m"].size for s in shards]
    assert min(sizes) > 0.8 * max(sizes)                        # the owner hash balances the shards


def test_mix_np_is_the_library_hash():
    """dist_build.mix_np restates qh_mix ("feistel3-core/2"): equal to the library's host hash, injective,
    invertible; the four end-base siblings of a mer share the low 2k-4 bits in both orientations; the
    all-ones item is free at k = 31"""
    import ctypes as C
    sys.path.insert(0, ROOT)
    from quorum_amd.dist_build import mix_np, unmix_np, pkey_np, key_from_pkey_np, hash_np, unhash_np
    from quorum_amd._lib import lib, check

    def lib_hash(x, k):
        x = np.ascontiguousarray(x, dtype=np.uint64)
        out = np.empty_like(x)
        u64p = C.POINTER(C.c_uint64)
        check(lib.qh_mer_hash(k, x.ctypes.data_as(u64p), x.size, out.ctypes.data_as(u64p)))
        return out

    for k in (1, 3, 4, 5, 8):
        x = np.arange(4 ** k, dtype=np.uint64)
        pk = pkey_np(x, k)
        assert np.unique(pk).size == x.size and (key_from_pkey_np(pk, k) == x).all()
        h = hash_np(x, k)
        assert (h == lib_hash(x, k)).all()
        assert np.unique(h).size == x.size and int(h.max()) < 2 * 4 ** k
        assert (unhash_np(h, k) == x).all()
    rng = np.random.default_rng(5)
    for k in (15, 24, 31):
        x = rng.integers(0, 4 ** k, size=20000, dtype=np.uint64)
        h = hash_np(x, k)
        assert (h == lib_hash(x, k)).all() and (unhash_np(h, k) == x).all()
        low = np.uint64((1 << (2 * k - 4)) - 1)
        rc = np.array([_revcomp(int(m), k) for m in x[:2000]], dtype=np.uint64)
        for b in range(4):
            sib0 = (x[:2000] & ~np.uint64(3)) | np.uint64(b)                      # last base varied
            sibk = (x[:2000] & ~(np.uint64(3) << np.uint64(2 * k - 2))) | (np.uint64(b) << np.uint64(2 * k - 2))
            assert ((hash_np(sib0, k) & low) == (h[:2000] & low)).all()
            assert ((hash_np(sibk, k) & low) == (h[:2000] & low)).all()
        assert ((hash_np(rc, k) & low) == (h[:2000] & low)).all()                 # and the reverse complement
    # the partition kernels mark "no item" with ~0 = (all-ones hash << 1 | 1): no 31-mer hashes to it
    ones = np.array([(1 << 63) - 1], dtype=np.uint64)
    assert hash_np(unhash_np(ones, 31), 31)[0] != ones[0]
    # ACGT = 00 01 10 11: low bits 0101 -> 0b0101, high bits 0011
    assert int(pkey_np(np.array([0x1B], dtype=np.uint64), 4)[0]) == 0b0011_0101


def _revcomp(m, k):
    r = 0
    for _ in range(k):
        r = (r << 2) | (3 - (m & 3))
        m >>= 2
    return r
