"""The no-allocation entry point qh_build_add_mers_device_noalloc against the entry point it is a variant of,
qh_build_add_mers_device, through the C ABI: same items -> bit-identical table content and counters, on two
calls in a row, the second with a smaller input (the library's scratch array keeps the first call's
contents). Collected last on purpose (the round's end runs pytest -x)."""
import ctypes as C

import numpy as np
import pytest
import torch

import quorum_amd as qa
from quorum_amd import _lib
from quorum_amd._lib import lib
from helpers import sorted_pairs, dict_model_np

pytestmark = pytest.mark.gpu

K, BITS, SIZE = 24, 7, 1 << 21


def make_items(rng, n, distinct):
    """n (mer << 1 | quality) items over `distinct` different mers, as the C ABI takes them"""
    pool = rng.integers(0, 1 << (2 * K), size=distinct, dtype=np.uint64)
    mers = pool[rng.integers(0, distinct, size=n)]
    q = rng.integers(0, 2, size=n, dtype=np.uint64)
    return mers, q, (mers << np.uint64(1)) | q


def test_noalloc_matches_add_mers_device_twice():
    rng = np.random.default_rng(77)
    ref = qa.HashWithQuality(SIZE, 2 * K, BITS, device=0)
    ref.set_mode("direct")           # the same kernel on both sides; content would match on any strategy
    na = qa.HashWithQuality(SIZE, 2 * K, BITS, device=0)
    all_m, all_q = [], []
    for n, distinct in ((400_000, 150_000), (30_000, 20_000)):   # the second call is smaller
        mers, q, items = make_items(rng, n, distinct)
        all_m.append(mers)
        all_q.append(q)
        dev = torch.from_numpy(items.view(np.int64)).to("cuda:0")
        ref.add_mers_device(dev.data_ptr(), n)
        ref.sync()
        # the property in the entry point's name: the driver's free-memory figure is the same before and after the
        # call (hipMemGetInfo; any hipMalloc on the path, e.g. an overflow list, would lower it)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(0)[0]
        na.add_mers_device_noalloc(dev.data_ptr(), n)   # returns when the items are in the table
        assert torch.cuda.mem_get_info(0)[0] == free_before
        rm, rv = sorted_pairs(*ref.dump())
        nm, nv = sorted_pairs(*na.dump())
        assert nm.size == rm.size and (nm == rm).all() and (nv == rv).all()
        rs, ns = ref.stats(), na.stats()
        for key in ("instances", "hq_instances", "distinct"):
            assert rs[key] == ns[key], (key, rs, ns)
        # and both against the (H, L) dictionary model of the update rule
        em, ev = dict_model_np(np.concatenate(all_m), np.concatenate(all_q).astype(np.uint8), BITS)
        assert (nm == em).all() and (nv == ev).all()
    ref.close()
    na.close()


def test_noalloc_rejects_oversize_and_pending():
    h = qa.HashWithQuality(1 << 16, 2 * K, BITS, device=0)
    dev = torch.zeros(16, dtype=torch.int64, device="cuda:0")
    st = lib.qh_build_add_mers_device_noalloc(h._h, C.c_void_p(dev.data_ptr()), (1 << 24) + 1)
    assert st == _lib.QH_EBADARG and "at most" in _lib.last_error()
    # a chunk of another entry point in flight must be collected first
    h.add_mers_device(dev.data_ptr(), 16)
    st = lib.qh_build_add_mers_device_noalloc(h._h, C.c_void_p(dev.data_ptr()), 16)
    assert st == _lib.QH_EBADARG and "qh_build_sync" in _lib.last_error()
    h.sync()
    h.add_mers_device_noalloc(dev.data_ptr(), 16)
    m, v = h.dump()
    assert m.size == 1 and int(m[0]) == 0 and int(v[0]) == (32 << 1)   # 32 low-quality sightings of AAAA...A
    h.close()


def test_noalloc_full_table_is_an_error():
    # 64 slots cannot hold 1000 distinct mers and the entry point may not grow the table
    h = qa.HashWithQuality(64, 2 * K, BITS, device=0)
    items = (np.arange(1, 1001, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15 >> 20)) & np.uint64((1 << 48) - 1)
    dev = torch.from_numpy((items << np.uint64(1)).view(np.int64)).to("cuda:0")
    st = lib.qh_build_add_mers_device_noalloc(h._h, C.c_void_p(dev.data_ptr()), 1000)
    assert st == _lib.QH_EFULL and "Hash is full" in _lib.last_error()
    h.close()
