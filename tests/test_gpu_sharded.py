"""The hash-sharded build on one GPU: 2^b shard handles in one process play the b-GPU
ranks (the RCCL collective itself is covered by tests/test_dist_cpu.py over gloo).
Checks bucketing by owner, per-shard insertion, shard growth, and that the shards
serialise into ONE database file that database_query reads back bit-exactly."""
import ctypes as C

import numpy as np
import pytest
import torch

import pyoracle as po
import quorum_amd as qa
from quorum_amd import _lib, synth
from quorum_amd.dist_build import owner_np, unhash_np
from helpers import sorted_pairs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shard_bits,bits", [(1, 7), (2, 6), (3, 7), (3, 2)])
def test_sharded_build_single_process(shard_bits, bits, tmp_path):
    # bits 6 and 2: value fields of 7 and 3 bits do not pack evenly into 64-bit words, so neighbouring shards share
    # boundary words of the values image (completed by read-OR-write, one writer at a time)
    k, thresh = 24, 38
    n = 1 << shard_bits
    g = synth.make_genome(200_000)
    fq = synth.make_fastq(g, 40_000, 100, tails=True)
    chunk = torch.from_numpy(fq).to("cuda:0")
    total_size = 1 << 21
    hs = [qa.HashWithQuality(total_size, 2 * k, bits, min_qual=thresh, shard_bits=shard_bits, shard_id=s) for s in range(n)]
    cap = fq.size // 2
    buckets = torch.zeros((n, cap), dtype=torch.int64, device="cuda:0")
    counts = torch.zeros(n, dtype=torch.int64, device="cuda:0")
    _lib.check(_lib.lib.qh_build_extract_partitioned_device(hs[0]._h, C.c_void_p(chunk.data_ptr()), chunk.numel(),
                                                            C.c_void_p(buckets.data_ptr()), cap, C.c_void_p(counts.data_ptr())))
    cnt = counts.tolist()
    db = po.OracleDB(total_size, k, bits)
    ninst = db.add_fastq(fq.tobytes(), thresh, threads=8)
    assert sum(cnt) == ninst
    for s in range(n):
        items = buckets[s, :cnt[s]].cpu().numpy().view(np.uint64)
        hashes = items >> np.uint64(1)
        assert ((hashes & np.uint64(n - 1)) == s).all()                      # owner = low shard_bits of the hash
        assert (owner_np(unhash_np(hashes, k), k, shard_bits) == s).all()    # python restatement == device hash
        hs[s].add_hashed_device(buckets[s].data_ptr(), cnt[s])
        hs[s].sync()
    om, ov = db.sorted_dump()
    dumps = [h.dump() for h in hs]
    m = np.concatenate([d[0] for d in dumps])
    v = np.concatenate([d[1] for d in dumps])
    assert np.unique(m).size == m.size
    sm, sv = sorted_pairs(m, v)
    assert (sm == om).all() and (sv == ov).all()
    # the no-exchange route: every shard scans the whole chunk and keeps what it owns
    for s in (0, n - 1):
        h2 = qa.HashWithQuality(total_size, 2 * k, bits, min_qual=thresh, shard_bits=shard_bits, shard_id=s)
        h2.add_fastq_device(chunk.data_ptr(), chunk.numel())
        st = h2.done()
        assert st["instances"] == cnt[s] and st["instances"] + st["foreign"] == ninst
        a, b = sorted_pairs(*h2.dump())
        c, d = sorted_pairs(*dumps[s])
        assert (a == c).all() and (b == d).all()
        h2.close()
    # one shard doubles (as after an overflow); the others are equalised before serialisation
    lsize = int(hs[0].stats()["slots"]).bit_length() - 1
    for h in hs:
        _lib.check(_lib.lib.qh_build_grow_to(h._h, lsize + 1))
        assert h.stats()["slots"] == 1 << (lsize + 1)
    path = str(tmp_path / "sharded.db")
    for s, h in enumerate(hs):
        _lib.check(_lib.lib.qh_build_write_shard(h._h, path.encode(), 0, None, 1 if s == 0 else 0))
    q = qa.DatabaseQuery(path)
    assert q.size == n << (lsize + 1)
    qm, qv = sorted_pairs(*q.dump())
    assert (qm == om).all() and (qv == ov).all()
    assert (q.lookup(om[::7]) == ov[::7]).all()
    assert q.stats() == db.stats()
    q.close()
    # the values image as compute_poisson_cutoff__ scans it (atomic_bits_array layout over ALL slots of the file)
    info = _lib.QhDbInfo()
    _lib.check(_lib.lib.qh_db_header_info(path.encode(), C.byref(info)))
    raw = np.fromfile(path, dtype=np.uint64, offset=info.offset + info.key_bytes)
    vpw, vb = 64 // (bits + 1), bits + 1
    fields = (raw[:, None] >> (np.arange(vpw, dtype=np.uint64) * np.uint64(vb))[None, :]) & np.uint64((1 << vb) - 1)
    fields = fields.reshape(-1)[:info.size]
    assert int((fields != 0).sum()) == om.size and np.array_equal(np.sort(fields[fields != 0]), np.sort(ov))
    for h in hs:
        h.close()


def test_two_rank_bench_rehearsal():
    """bench.py's N>1 path end to end (BASELINE configs[2] shape: reads drawn on the GPU, pieces through the
    extract / exchange / insert pipeline) with two ranks sharing the one GPU and gloo standing in for RCCL."""
    import json
    import os
    import subprocess
    import sys
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, QH_DIST_BACKEND="gloo", QH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "200000", "--piece-reads", "70000"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "strong" and "configs[2]" in d["config"]["workload"]
    # 2 ranks x 200k reads x (100-24+1) windows, minus the windows that contain an N
    assert 0.97 * 2 * 200000 * 77 < d["config"]["insertions_per_step"] <= 2 * 200000 * 77
    assert d["config"]["pieces_per_gpu"] == 3
    # the run checked itself: both shards' content against a single-table build of the same reads
    assert d["content_check"]["ok"] is True and d["content_check"]["reads_per_rank"] == 60000
    assert d["exchange_message_cap"] == 8 << 26 and "large_message_ok" not in d   # (the large message is RCCL's test)


def _pipeline_rank(rank, world, port, outdir):
    """one rank of the pipelined sharded build: gloo between the ranks, both on GPU 0"""
    import os
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch_model as tm
    from quorum_amd.dist_build import ShardedBuilder
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        k, bits, thresh, L = 24, 7, 38, 100
        genome = tm.make_genome(300_000, "cuda:0")
        # rank 0 feeds three pieces, rank 1 two and then None; the small bucket capacity forces the overflow retry
        b = ShardedBuilder(k, bits, 1 << 23, thresh, rank, world, 0, cap_factor=0.4)
        n_pieces = 3 if rank == 0 else 2
        canon, hq = [], []
        for i in range(3):
            if i < n_pieces:
                codes, qual, isN = tm.make_reads(genome, 30_000 + 5_000 * i, L, seed=10 * rank + i, tails=True)
                chunk = tm.fastq_text(codes, qual, isN)
                c, q = tm.model_items(codes, qual, isN, k, thresh)
                canon.append(c.cpu())
                hq.append(q.cpu())
                torch.cuda.synchronize()
                b.add_fastq_device(chunk)
            else:
                b.add_fastq_device(None)
        st = b.finish()
        m, v = b.h.dump()
        np.savez(os.path.join(outdir, "rank%d.npz" % rank), m=m, v=v, canon=torch.cat(canon).numpy(), hq=torch.cat(hq).numpy(),
                 inst=np.array([st["instances"]]), retries=np.array([b.retries]))
    finally:
        dist.destroy_process_group()


def test_pipelined_sharded_builder_two_ranks(tmp_path):
    """ShardedBuilder's extract / exchange / insert pipeline with two ranks (gloo, sharing the GPU): uneven numbers of
    pieces, bucket overflow resolved by the retry, union of the shards == the model of all reads"""
    import socket
    import torch.multiprocessing as mp
    import torch_model as tm
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    mp.start_processes(_pipeline_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r = [np.load(str(tmp_path / ("rank%d.npz" % i))) for i in range(2)]
    canon = torch.from_numpy(np.concatenate([x["canon"] for x in r]))
    hq = torch.from_numpy(np.concatenate([x["hq"] for x in r]))
    mm, mv = tm.model_table(canon, hq, 7)
    assert sum(int(x["inst"][0]) for x in r) == canon.numel()
    assert all(int(x["retries"][0]) >= 1 for x in r)
    m = np.concatenate([x["m"] for x in r])
    v = np.concatenate([x["v"] for x in r])
    assert np.unique(m).size == m.size
    sm, sv = sorted_pairs(m, v)
    assert (sm == mm.numpy().view(np.uint64)).all() and (sv == mv.numpy().view(np.uint64)).all()
    assert (owner_np(r[0]["m"], 24, 1) == 0).all() and (owner_np(r[1]["m"], 24, 1) == 1).all()


def _rccl_self_rank(rank, world, port, outdir):
    """the one rank of test_pipeline_over_rccl_single_rank"""
    import os
    import sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch_model as tm
    os.environ["QH_EXCHANGE_MAX_ITEMS"] = "600000"     # a piece of ~2.5 M items travels in several capped rounds
    from quorum_amd.dist_build import ShardedBuilder, exchange_buckets, exchange_items
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    try:
        k, bits, thresh, L = 24, 7, 38, 100
        genome = tm.make_genome(300_000, "cuda:0")
        b = ShardedBuilder(k, bits, 1 << 23, thresh, 0, 1, 0, cap_factor=0.4, force_pipeline=True)
        canon, hq = [], []
        for i in range(4):
            if i == 2:
                b.add_fastq_device(None)       # a piece with nothing in it travels too
                continue
            codes, qual, isN = tm.make_reads(genome, 30_000 + 5_000 * i, L, seed=i, tails=True)
            chunk = tm.fastq_text(codes, qual, isN)
            c, q = tm.model_items(codes, qual, isN, k, thresh)
            canon.append(c.cpu())
            hq.append(q.cpu())
            torch.cuda.synchronize()
            b.add_fastq_device(chunk)
        st = b.finish()
        m, v = b.h.dump()
        # the two stand-alone exchange helpers over RCCL as well
        rows = torch.arange(40, dtype=torch.int64, device=dev).reshape(1, 40)
        got, rc = exchange_buckets(rows, [17])
        got2, rc2 = exchange_items(rows[0, :9].contiguous(), [9])
        torch.cuda.synchronize()
        np.savez(os.path.join(outdir, "rccl.npz"), m=m, v=v, canon=torch.cat(canon).numpy(), hq=torch.cat(hq).numpy(),
                 inst=np.array([st["instances"]]), retries=np.array([b.retries, b.rounds_max]), got=got.cpu().numpy(),
                 got2=got2.cpu().numpy(), rc=np.array(rc + rc2))
    finally:
        dist.destroy_process_group()


def test_pipeline_over_rccl_single_rank(tmp_path):
    """The RCCL calls of the sharded build on the hardware there is: one rank, backend "nccl", the pipeline forced
    (extract by owner -> counts ahead -> list-form all-to-all started asynchronously -> insert on the shard handle's stream
    after work.wait()), the message cap lowered so that every piece takes several rounds. The collective sends to itself,
    but it is the same torch.distributed / RCCL call sequence the N > 1 bench issues; content against the independent
    model."""
    import socket
    import torch.multiprocessing as mp
    import torch_model as tm
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    mp.start_processes(_rccl_self_rank, args=(1, port, str(tmp_path)), nprocs=1, join=True, start_method="spawn")
    r = np.load(str(tmp_path / "rccl.npz"))
    mm, mv = tm.model_table(torch.from_numpy(r["canon"]), torch.from_numpy(r["hq"]), 7)
    assert int(r["inst"][0]) == r["canon"].size and int(r["retries"][0]) >= 1 and int(r["retries"][1]) >= 3
    sm, sv = sorted_pairs(r["m"], r["v"])
    assert (sm == mm.numpy().view(np.uint64)).all() and (sv == mv.numpy().view(np.uint64)).all()
    assert (r["got"] == np.arange(17)).all() and (r["got2"] == np.arange(9)).all() and list(r["rc"]) == [17, 9]


def _multi_rccl_child(outdir):
    """qh_multi with ONE shard and the RCCL transport forced (own process: the transport is chosen from the environment)"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["QH_MULTI_TRANSPORT"] = "rccl"
    os.environ["QH_EXCHANGE_MAX_ITEMS"] = "300000"
    import torch  # noqa: F401  (brings librccl.so into the process the way the tools find it)
    from quorum_amd import _lib, synth as sy
    k, bits = 24, 7
    g = sy.make_genome(150000)
    fq = sy.make_fastq(g, 40000, 100, tails=True).tobytes()
    h = C.c_void_p()
    _lib.check(_lib.lib.qh_multi_create(C.byref(h), k, bits, 1 << 22, 38, 126, 1, None))
    used = C.c_size_t()
    half = fq.index(b"\n@", len(fq) // 2) + 1
    _lib.check(_lib.lib.qh_multi_add_fastq(h, fq[:half], half, 0, C.byref(used)))
    assert used.value == half
    _lib.check(_lib.lib.qh_multi_add_fastq(h, fq[half:], len(fq) - half, 1, C.byref(used)))
    st = _lib.QhBuildStats()
    _lib.check(_lib.lib.qh_multi_finish(h, C.byref(st)))
    nd, rccl, rounds = C.c_int(), C.c_int(), C.c_uint64()
    _lib.check(_lib.lib.qh_multi_info(h, C.byref(nd), C.byref(rccl), C.byref(rounds), None, None))
    cap = int(st.distinct) + 16
    m = np.empty(cap, dtype=np.uint64)
    v = np.empty(cap, dtype=np.uint64)
    n = C.c_uint64()
    u64p = C.POINTER(C.c_uint64)
    _lib.check(_lib.lib.qh_multi_dump(h, m.ctypes.data_as(u64p), v.ctypes.data_as(u64p), cap, C.byref(n)))
    _lib.lib.qh_multi_destroy(h)
    np.savez(os.path.join(outdir, "multi.npz"), m=m[:n.value], v=v[:n.value], fq=np.frombuffer(fq, dtype=np.uint8),
             info=np.array([nd.value, rccl.value, rounds.value, st.instances]))


def test_multi_host_over_rccl_single_shard(tmp_path):
    """The C++ multi-GPU host's RCCL leg on the hardware there is: librccl through dlopen, ncclCommInitAll, the grouped
    ncclSend / ncclRecv of every chunk in capped rounds (cap lowered: a chunk of ~1.5 M items takes five), here with one
    shard sending to itself; content against the oracle."""
    import torch.multiprocessing as mp
    mp.start_processes(_multi_rccl_child_entry, args=(str(tmp_path),), nprocs=1, join=True, start_method="spawn")
    r = np.load(str(tmp_path / "multi.npz"))
    assert list(r["info"][:2]) == [1, 1] and int(r["info"][2]) == 2
    fq = r["fq"].tobytes()
    db = po.OracleDB(1 << 22, 24, 7)
    n = db.add_fastq(fq, 38, threads=8)
    assert n == int(r["info"][3])
    om, ov = db.sorted_dump()
    m, v = sorted_pairs(r["m"], r["v"])
    assert m.size == om.size and (m == om).all() and (v == ov).all()


def _multi_rccl_child_entry(rank, outdir):
    _multi_rccl_child(outdir)
