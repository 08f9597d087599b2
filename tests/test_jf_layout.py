"""The isolated jellyfish-layout keys image (quorum_amd/csrc/qh_jflayout.h): host-only self round trip.

PARITY UNPINNED: nothing here is checked against Jellyfish (absent from the reference tree and from this image);
the tests pin the module against itself and against the format rules SURVEY.md 8(c) records -- entry width, image
size, header keys, quadratic reprobes, value fields of the atomic_bits_array -- so that the one place to fix is
known once an upstream reader is at hand.
"""
import json

import numpy as np
import pytest

from quorum_amd import _lib
from quorum_amd import mer_database as md


def _pairs(k, bits, n, seed):
    rng = np.random.default_rng(seed)
    mers = np.unique(rng.integers(0, 1 << (2 * k), size=2 * n, dtype=np.uint64))
    mers = rng.permutation(mers)[:n]
    vals = rng.integers(1, 1 << (bits + 1), size=mers.size, dtype=np.uint64)
    return mers, vals


def _header(path):
    raw = open(path, "rb").read()
    hlen = int(raw[:9])
    return json.loads(raw[9:9 + hlen].rstrip(b"\0")), 9 + hlen, len(raw)


@pytest.mark.parametrize("k,bits,size,n", [(24, 7, 1 << 16, 40000), (31, 7, 1 << 12, 3000), (15, 3, 1 << 10, 700),
                                           (17, 6, 1 << 14, 1 << 14), (24, 7, 1 << 10, 0)])
def test_round_trip(tmp_path, k, bits, size, n):
    mers, vals = _pairs(k, bits, n, seed=k * 1000 + bits)
    path = tmp_path / "db.jf"
    md.jf_write_file(path, k, bits, size, mers, vals, argv=["prog", "-s", str(size)])
    k2, b2, size2, m2, v2 = md.jf_read_file(path)
    assert (k2, b2) == (k, bits) and size2 >= size and size2 & (size2 - 1) == 0
    got = dict(zip(m2.tolist(), v2.tolist()))
    assert len(got) == m2.size == mers.size
    assert got == dict(zip(mers.tolist(), vals.tolist()))

    hdr, offset, flen = _header(path)
    assert "qhip" not in hdr and hdr["format"] == "binary/quorum_db"
    lsize = size2.bit_length() - 1
    assert hdr["size"] == size2 and hdr["key_len"] == 2 * k and hdr["val_len"] == 0 and hdr["bits"] == bits
    assert hdr["matrix1"]["r"] == lsize and hdr["matrix1"]["c"] == 2 * k and len(hdr["matrix1"]["columns"]) == 2 * k
    assert all(0 <= c < size2 for c in hdr["matrix1"]["columns"])
    mr = hdr["max_reprobe"]
    assert hdr["reprobes"] == [i * (i + 1) // 2 for i in range(mr + 1)] and hdr["reprobes"][-1] < size2
    # 1 large bit + stored key bits + reprobe field, packed back to back into whole words
    entry_bits = 1 + 2 * k - lsize + (mr + 1).bit_length()
    assert hdr["key_bytes"] == (size2 * entry_bits + 63) // 64 * 8
    vpw = 64 // (bits + 1)
    assert hdr["value_bytes"] == (size2 + vpw - 1) // vpw * 8
    # the invariants the reference's test holds for the container (unit_tests/test_mer_database.cc:102-106)
    assert offset % 8 == 0 and flen == offset + hdr["key_bytes"] + hdr["value_bytes"]
    assert hdr["cmdline"] == ["prog", "-s", str(size)]


def test_entries_sit_at_matrix_position_plus_reprobe(tmp_path):
    """decode the image independently in numpy: entry id = (M x key + reprobes[r]) mod size, field layout as recorded"""
    k, bits, size, n = 20, 7, 1 << 12, 3000
    mers, vals = _pairs(k, bits, n, seed=5)
    path = tmp_path / "db.jf"
    md.jf_write_file(path, k, bits, size, mers, vals)
    hdr, offset, _ = _header(path)
    raw = open(path, "rb").read()
    lsize = hdr["size"].bit_length() - 1
    cols = hdr["matrix1"]["columns"]
    rl = (hdr["max_reprobe"] + 1).bit_length()
    eb = 1 + 2 * k - lsize + rl
    image = int.from_bytes(raw[offset:offset + hdr["key_bytes"]], "little")
    vimg = np.frombuffer(raw, dtype="<u8", offset=offset + hdr["key_bytes"])
    vpw = 64 // (bits + 1)
    want = dict(zip(mers.tolist(), vals.tolist()))
    seen = 0
    for eid in range(hdr["size"]):
        e = (image >> (eid * eb)) & ((1 << eb) - 1)
        if not e:
            continue
        assert e & 1 == 0
        f = e >> 1
        rp, high = (f & ((1 << rl) - 1)) - 1, f >> rl
        # the key is the one among the expected keys with these high bits whose position matches
        pos = (eid - hdr["reprobes"][rp]) % hdr["size"]
        hit = None
        for key in want:
            if key >> lsize != high:
                continue
            p = 0
            for i in range(2 * k):
                if (key >> i) & 1:
                    p ^= cols[2 * k - 1 - i]
            if p == pos:
                assert hit is None
                hit = key
        assert hit is not None
        v = (int(vimg[eid // vpw]) >> ((eid % vpw) * (bits + 1))) & ((1 << (bits + 1)) - 1)
        assert v == want[hit]
        seen += 1
    assert seen == n


def test_rejects_what_does_not_fit(tmp_path):
    with pytest.raises(_lib.QhError):  # 2k <= log2(size): nothing left to store
        md.jf_write_file(tmp_path / "x", 5, 7, 1 << 12, np.array([1], dtype=np.uint64), np.array([3], dtype=np.uint64))
    with pytest.raises(_lib.QhError):  # a value of 0 means "empty"
        md.jf_write_file(tmp_path / "x", 24, 7, 1 << 12, np.array([1], dtype=np.uint64), np.array([0], dtype=np.uint64))
    with pytest.raises(_lib.QhError):  # value wider than bits + 1
        md.jf_write_file(tmp_path / "x", 24, 3, 1 << 12, np.array([1], dtype=np.uint64), np.array([16], dtype=np.uint64))
    p = tmp_path / "native_like"
    p.write_bytes(b"000000016" + b'{"format":"x"}\n\0')
    with pytest.raises(_lib.QhError):
        md.jf_read_file(p)


def test_grows_when_reprobe_limit_is_hit(tmp_path):
    k, bits = 24, 7
    mers, vals = _pairs(k, bits, 1000, seed=9)
    path = tmp_path / "db.jf"
    md.jf_write_file(path, k, bits, 1024, mers, vals, reprobe_limit=2)  # 98 % load, 3 probes: cannot fit
    _, _, size2, m2, v2 = md.jf_read_file(path)
    assert size2 > 1024 and dict(zip(m2.tolist(), v2.tolist())) == dict(zip(mers.tolist(), vals.tolist()))


def test_host_module_under_sanitizers(tmp_path):
    """the module's C++ (bit packing, GF(2) inverse, file reader) built with g++ -fsanitize=address,undefined and run on the
    CPU: round trips, growth, rejected inputs (tests/cpp/test_jflayout_host.cpp)"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "quorum_amd", "csrc")
    exe = str(tmp_path / "jf_host")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
           os.path.join(root, "tests", "cpp", "test_jflayout_host.cpp"), os.path.join(csrc, "qh_jflayout.cpp"),
           os.path.join(csrc, "qh_dbfile.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.stdout + r.stderr)[-3000:]
