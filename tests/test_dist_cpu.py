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


def _worker(rank, world, port, k, bits, outdir):
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
        gm, gq = got >> np.uint64(1), (got & np.uint64(1)).astype(np.uint8)
        assert (owner_np(gm, k, shard_bits) == rank).all()      # only what this shard owns arrives
        db = po.OracleDB(1 << 16, k, bits)
        db.add_mers(gm, gq)
        m, v = db.sorted_dump()
        np.savez(os.path.join(outdir, "shard%d.npz" % rank), m=m, v=v, n=np.array([got.size]), sent=np.array([items.size]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_exchange_gloo(world, tmp_path):
    k, bits = 21, 7
    port = _free_port()
    mp.spawn(_worker, args=(world, port, k, bits, str(tmp_path)), nprocs=world, join=True)
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
    """dist_build.mix_np restates qh_mix; check bijectivity and the pinned constants"""
    sys.path.insert(0, ROOT)
    from quorum_amd.dist_build import mix_np, unmix_np, pkey_np, key_from_pkey_np, hash_np, unhash_np
    for k in (4, 5, 8):
        x = np.arange(4 ** k, dtype=np.uint64)
        h = mix_np(x, k)
        assert np.unique(h).size == x.size and int(h.max()) < 4 ** k
        assert (unmix_np(h, k) == x).all()
        pk = pkey_np(x, k)
        assert np.unique(pk).size == x.size and (key_from_pkey_np(pk, k) == x).all()
        assert (unhash_np(hash_np(x, k), k) == x).all()
    # ACGT = 00 01 10 11: low bits 0101 -> 0b0101, high bits 0011
    assert int(pkey_np(np.array([0x1B], dtype=np.uint64), 4)[0]) == 0b0011_0101
    assert mix_np(np.array([0x123456789ABC], dtype=np.uint64), 24)[0] == mix_np(np.array([0x123456789ABC], dtype=np.uint64), 24)[0]
