"""The reference's own unit test (unit_tests/test_mer_database.cc:47-139, TEST_P WriteRead with
INSTANTIATE Values(1, 10, 20, 40)) compiled against the header-only C++ shim
include/quorum_hip/mer_database.hpp and run on the GPU. Source: tests/cpp/test_mer_database_shim.cpp,
built by quorum_amd/csrc/Makefile into quorum_amd/bin/."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

BIN = os.path.join(os.path.dirname(__file__), "..", "quorum_amd", "bin", "test_mer_database_shim")


@pytest.mark.parametrize("param", [1, 10, 20, 40])
def test_write_read(param, tmp_path):
    assert os.path.exists(BIN), "run __graft_entry__.build() first"
    r = subprocess.run([BIN, str(param), str(tmp_path / "db.jf")], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("PASSED")
