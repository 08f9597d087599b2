"""Pins the whole-tensor model of tests/torch_model.py (the checker of the full-size GPU tests) to the oracle:
same reads -> same (mer, count << 1 | hq) table. Runs on the CPU, and once more on the GPU (marked gpu) so that the
driver's GPU run checks the checker of tests/test_gpu_big_configs.py on the device it is used on."""
import numpy as np
import pytest
import torch

import pyoracle as po
import torch_model as tm


@pytest.mark.parametrize("k,L,bits,thresh,tails", [(24, 100, 7, 38, False), (31, 150, 7, 38, True), (5, 40, 2, 50, True),
                                                   (17, 60, 1, 38, False)])
def test_torch_model_matches_oracle(k, L, bits, thresh, tails):
    check_model(k, L, bits, thresh, tails, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("k,L,bits,thresh,tails", [(24, 100, 7, 38, False), (31, 150, 7, 38, True), (5, 40, 2, 50, True)])
def test_torch_model_matches_oracle_on_the_gpu(k, L, bits, thresh, tails):
    check_model(k, L, bits, thresh, tails, "cuda", reads=6000)


def check_model(k, L, bits, thresh, tails, device, reads=1500):
    genome = tm.make_genome(3000, device, seed=k)
    codes, qual, isN = tm.make_reads(genome, reads, L, seed=7 * k, err=0.02, n_rate=0.004, tails=tails)
    text = tm.fastq_text(codes, qual, isN).cpu().numpy().tobytes()
    db = po.OracleDB(1 << 18, k, bits)
    n = db.add_fastq(text, thresh)
    om, ov = db.sorted_dump()
    c, q = tm.model_items(codes, qual, isN, k, thresh)
    assert c.numel() == n
    mm, mv = tm.model_table(c, q, bits)
    mm, mv = mm.cpu(), mv.cpu()
    assert mm.numel() == om.size
    assert (mm.numpy().view(np.uint64) == om).all() and (mv.numpy().view(np.uint64) == ov).all()
    assert int((mv & 1).sum()) > 0 and int((mv & 1).sum()) < mv.numel()   # both quality classes occur
