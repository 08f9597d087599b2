"""CPU-only checks of the product's host side: the C-ABI library loads and exports
every symbol include/quorum_hip.h declares, argument validation happens before any
GPU call, the CLI rejects what the reference rejects, header parsing works without a
GPU. No compute calls here (no GPU)."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

import quorum_amd
from quorum_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "quorum_amd", "bin", "quorum_create_database")


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "quorum_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(_lib.lib, n), "libquorum_hip.so does not export %s" % n
    bound = {s[0] for s in _lib.SIGNATURES}
    assert set(names) <= bound, "python binding misses %s" % (set(names) - bound)


def test_version_and_error_string():
    assert b"quorum_hip" in _lib.lib.qh_version()
    assert isinstance(quorum_amd.last_error(), str)


@pytest.mark.parametrize("k,bits,size,msg", [
    (0, 7, 100, "Mer length"), (32, 7, 100, "Mer length"),
    (24, 0, 100, "bits should be between 1 and 63"), (24, 64, 100, "bits should be between 1 and 63"),
    (24, 7, 0, "size")])
def test_create_rejects_bad_geometry_without_gpu(k, bits, size, msg):
    h = C.c_void_p()
    st = _lib.lib.qh_build_create(C.byref(h), k, bits, size, 38, 126, 0, 0, 0)
    assert st == _lib.QH_EBADARG
    assert msg in quorum_amd.last_error()


def test_null_handles_are_rejected():
    assert _lib.lib.qh_build_sync(None) == _lib.QH_EBADARG
    assert _lib.lib.qh_build_add_fastq(None, b"x", 1) == _lib.QH_EBADARG
    assert _lib.lib.qh_query_batch(None, None, 0, None) == _lib.QH_EBADARG
    _lib.lib.qh_build_destroy(None)
    _lib.lib.qh_query_close(None)


def test_fastq_4line_prefix():
    f = _lib.lib.qh_fastq_4line_prefix
    rec = b"@a\nACGT\n+\nIIII\n"
    assert f(rec * 3, len(rec) * 3, 0) == len(rec) * 3
    assert f(rec * 3 + b"@b\nAC", len(rec) * 3 + 5, 0) == len(rec) * 3
    assert f(rec[:-1], len(rec) - 1, 0) == 0
    assert f(rec[:-1], len(rec) - 1, 1) == len(rec) - 1  # final chunk: last line may lack '\n'
    assert f(b"", 0, 1) == 0


def write_db_file(path, header, payload=b""):
    j = json.dumps(header).encode()
    pad = (8 - (9 + len(j)) % 8) % 8
    with open(path, "wb") as f:
        f.write(b"%09d" % (len(j) + pad) + j + b"\0" * pad + payload)
    return 9 + len(j) + pad


def test_header_info_parses_reference_style_header(tmp_path):
    p = str(tmp_path / "db")
    hdr = {"format": "binary/quorum_db", "size": 1024, "key_len": 48, "val_len": 0, "max_reprobe": 126,
           "bits": 7, "key_bytes": 8192, "value_bytes": 1024, "cmdline": ["x", "-s", "1k"],
           "qhip": {"layout": "slot64/1", "hash": "feistel3-core/2", "window_bits": 10, "probe_bits": 7, "shard_bits": 0}}
    off = write_db_file(p, hdr, b"\0" * (8192 + 1024))
    info = _lib.QhDbInfo()
    assert _lib.lib.qh_db_header_info(p.encode(), C.byref(info)) == _lib.QH_OK
    assert (info.size, info.key_len, info.bits, info.key_bytes, info.value_bytes) == (1024, 48, 7, 8192, 1024)
    assert info.offset == off and off % 8 == 0  # test_mer_database.cc:102
    assert info.format == b"binary/quorum_db"


def test_header_info_errors(tmp_path):
    info = _lib.QhDbInfo()
    assert _lib.lib.qh_db_header_info(str(tmp_path / "missing").encode(), C.byref(info)) == _lib.QH_EIO
    assert "Can't open" in quorum_amd.last_error()  # mer_database.hpp:259-260
    p = str(tmp_path / "garbage")
    open(p, "wb").write(b"not a database at all")
    assert _lib.lib.qh_db_header_info(p.encode(), C.byref(info)) == _lib.QH_EIO
    assert "Can't parse header" in quorum_amd.last_error()  # :262-263
    p = str(tmp_path / "other")
    write_db_file(p, {"format": "binary/sorted", "size": 16})
    assert _lib.lib.qh_db_header_info(p.encode(), C.byref(info)) == _lib.QH_EFORMAT
    assert "Wrong type" in quorum_amd.last_error()  # :264-265


def run_cli(*args):
    return subprocess.run([CLI] + list(args), capture_output=True, text=True, timeout=60)


@pytest.mark.parametrize("args,msg", [
    (["-m", "24", "-b", "7", "-q", "38", "r.fq"], "[-s, --size=uint64] required switch"),
    (["-s", "1k", "-b", "7", "-q", "38", "r.fq"], "[-m, --mer=uint32] required switch"),
    (["-s", "1k", "-m", "24", "-q", "38", "r.fq"], "[-b, --bits=uint32] required switch"),
    (["-s", "1k", "-m", "24", "-b", "7", "r.fq"], "Either a min-qual-value or min-qual-char must be provided."),
    (["-s", "1k", "-m", "24", "-b", "7", "-Q", "ab", "r.fq"], "The min-qual-char should be one ASCII character."),
    (["-s", "1k", "-m", "24", "-b", "7", "-q", "38", "-Q", "a", "r.fq"], "mutually exclusive"),
    (["-s", "1k", "-m", "24", "-b", "0", "-q", "38", "r.fq"], "The number of bits should be between 1 and 63"),
    (["-s", "1k", "-m", "24", "-b", "64", "-q", "38", "r.fq"], "The number of bits should be between 1 and 63"),
    (["-s", "1k", "-m", "24", "-b", "7", "-q", "38"], "Requires at least 1 argument."),
    (["-s", "12x", "-m", "24", "-b", "7", "-q", "38", "r.fq"], "Invalid conversion"),
    (["-s", "1k", "-m", "24", "-b", "7", "-q", "38", "-o", "/nonexistent_dir/x", "r.fq"], "Failed to open output file"),
])
def test_cli_validation_matches_reference(args, msg, tmp_path):
    """create_database.cc:104-115 and the yaggo switch rules (create_database_cmdline.yaggo:7-33):
    message on stderr, exit status 1, before any work is done."""
    if "-o" not in args:
        args = ["-o", str(tmp_path / "out")] + args
    r = run_cli(*args)
    assert r.returncode == 1
    assert msg in r.stderr


def test_cli_help_lists_reference_switches():
    r = run_cli("--help")
    assert r.returncode == 0
    for sw in ("--size", "--mer", "--bits", "--min-qual-value", "--min-qual-char", "--threads", "--output", "--reprobe"):
        assert sw in r.stdout


def test_detect_min_qual_char_like_the_driver():
    """src/quorum.in:129-152 restated by hand: min over the quality lines of the first 1000 records,
    35 -> 33 and 66 -> 64, anything but 33 / 59 / 64 stops the run"""
    def fq(quals):
        return "".join("@r%d\nACGT\n+\n%s\n" % (i, q) for i, q in enumerate(quals)).encode()
    assert quorum_amd.detect_min_qual_char(fq(["IIII", "I#II", "!III"])) == 33
    assert quorum_amd.detect_min_qual_char(fq(["IIII", "I#II"])) == 33       # 35 - 2
    assert quorum_amd.detect_min_qual_char(fq(["hhhh", "hBhh"])) == 64       # 66 - 2
    assert quorum_amd.detect_min_qual_char(fq(["hhhh", "h@hh"])) == 64
    assert quorum_amd.detect_min_qual_char(fq(["hhhh", "h;hh"])) == 59
    with pytest.raises(quorum_amd.QhError, match="unusual minimum quality char of 40"):
        quorum_amd.detect_min_qual_char(fq(["IIII", "I(II"]))
    # only the first 1000 records count: a '!' in record 1001 is not seen
    with pytest.raises(quorum_amd.QhError, match="unusual minimum quality char of 104"):
        quorum_amd.detect_min_qual_char(fq(["hhhh"] * 1000 + ["!hhh"]))
    assert quorum_amd.detect_min_qual_char(fq(["hhhh"] * 999 + ["@hhh"] + ["!hhh"])) == 64
    with pytest.raises(quorum_amd.QhError, match="Invalid fastq format"):
        quorum_amd.detect_min_qual_char(b"@r0\nACGT\n+\n")


def test_mate_pair_helpers(tmp_path):
    """merge_mate_pairs (src/merge_mate_pairs.cc:50-91) interleaves the even and the odd files as 4-line
    FASTQ ('*' qualities for FASTA records, multi-line records re-flowed); split_mate_pairs
    (src/split_mate_pairs.cc:15-33) deals two-line records alternately to <prefix>_1.fa / _2.fa"""
    import subprocess
    bindir = os.path.join(ROOT, "quorum_amd", "bin")
    a1 = b"@a1 x\nACGT\nAC\n+\nIIII\nII\n@a2\nGG\n+a2\nII\n"
    b1 = b">b1\nTTTT\nTT\n>b2 y\nCC\n"
    a2 = b"@a3\nA\n+\nI\n"
    b2 = b"@b3\nC\n+\n#\n"
    for n, d in (("a1", a1), ("b1", b1), ("a2", a2), ("b2", b2)):
        (tmp_path / n).write_bytes(d)
    r = subprocess.run([os.path.join(bindir, "merge_mate_pairs")] + [str(tmp_path / n) for n in ("a1", "b1", "a2", "b2")],
                       capture_output=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert r.stdout == (b"@a1 x\nACGTAC\n+\nIIIIII\n@b1\nTTTTTT\n+\n******\n@a2\nGG\n+\nII\n@b2 y\nCC\n+\n**\n"
                        b"@a3\nA\n+\nI\n@b3\nC\n+\n#\n")
    r = subprocess.run([os.path.join(bindir, "merge_mate_pairs"), str(tmp_path / "a1"), str(tmp_path / "a2")],
                       capture_output=True, timeout=60)
    assert r.returncode != 0 and b"Input files are not paired reads." in r.stderr
    r = subprocess.run([os.path.join(bindir, "merge_mate_pairs"), str(tmp_path / "a1"), str(tmp_path / "b1"), str(tmp_path / "a2")],
                       capture_output=True, timeout=60)
    assert r.returncode == 1 and b"Must give a even number files" in r.stderr
    fa = b">a1 x 3:sub:A-C \nACCTAC\n>b1\nN\n>a2  \nGG\n>b2 y  \nCC\n"
    r = subprocess.run([os.path.join(bindir, "split_mate_pairs"), str(tmp_path / "out")], input=fa, capture_output=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "out_1.fa").read_bytes() == b">a1 x 3:sub:A-C \nACCTAC\n>a2  \nGG\n"
    assert (tmp_path / "out_2.fa").read_bytes() == b">b1\nN\n>b2 y  \nCC\n"


def test_mate_pair_helpers_edges(tmp_path):
    """the block readers of both helpers at their edges: last lines without a newline, a lone header at the end of the
    corrector's stream, lines longer than a first guess, a file that ends inside a record, the quality-count error"""
    import subprocess
    bindir = os.path.join(ROOT, "quorum_amd", "bin")
    split = os.path.join(bindir, "split_mate_pairs")
    merge = os.path.join(bindir, "merge_mate_pairs")
    for fa, one, two in ((b">a\nAC\n>b\nGT", b">a\nAC\n", b">b\nGT\n"),
                         (b">a\nAC\n>b\n", b">a\nAC\n", b">b\n\n"),
                         (b">a\nAC\n>b", b">a\nAC\n", b">b\n\n"),
                         (b"", b"", b"")):
        r = subprocess.run([split, str(tmp_path / "e")], input=fa, capture_output=True, timeout=60)
        assert r.returncode == 0, r.stderr
        assert (tmp_path / "e_1.fa").read_bytes() == one and (tmp_path / "e_2.fa").read_bytes() == two
    big = b"ACGT" * 3_000_000                      # 12 MB lines: longer than either tool's buffer
    fa = b">x\n" + big + b"\n>y\n" + big[:-1] + b"\n"
    r = subprocess.run([split, str(tmp_path / "l")], input=fa, capture_output=True, timeout=60)
    assert r.returncode == 0 and (tmp_path / "l_1.fa").read_bytes() == b">x\n" + big + b"\n"
    assert (tmp_path / "l_2.fa").read_bytes() == b">y\n" + big[:-1] + b"\n"
    (tmp_path / "m1").write_bytes(b"@x\n" + big + b"\n+\n" + b"I" * len(big))        # no final newline
    (tmp_path / "m2").write_bytes(b">y\n" + big[:100] + b"\n" + big[:50])
    r = subprocess.run([merge, str(tmp_path / "m1"), str(tmp_path / "m2")], capture_output=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert r.stdout == b"@x\n" + big + b"\n+\n" + b"I" * len(big) + b"\n@y\n" + big[:150] + b"\n+\n" + b"*" * 150 + b"\n"
    (tmp_path / "t1").write_bytes(b"@x\nACGT\n")
    (tmp_path / "t2").write_bytes(b"@y\nACGT\n+\nIII\n@z\nA\n+\nI\n")
    r = subprocess.run([merge, str(tmp_path / "t1"), str(tmp_path / "t1")], capture_output=True, timeout=60)
    assert r.returncode == 134 and b"Truncated fastq file" in r.stderr
    r = subprocess.run([merge, str(tmp_path / "t2"), str(tmp_path / "t2")], capture_output=True, timeout=60)
    assert r.returncode == 134 and b"wrong number of quals" in r.stderr
    r = subprocess.run([merge, str(tmp_path / "m2"), str(tmp_path / "missing")], capture_output=True, timeout=60)
    assert r.returncode == 134 and b"Can't open file" in r.stderr


def _model_parse(data):
    """the reader of merge_mate_pairs restated: FASTA / FASTQ, multi-line sequences and qualities re-flowed"""
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    out, i = [], 0
    while i < len(lines):
        if lines[i].startswith(b">"):
            h, i, seq = lines[i][1:], i + 1, b""
            while i < len(lines) and not lines[i].startswith(b">"):
                seq += lines[i]
                i += 1
            out.append((h, seq, None))
        else:
            assert lines[i].startswith(b"@")
            h, i, seq, qual = lines[i][1:], i + 1, b"", b""
            while i < len(lines) and not lines[i].startswith(b"+"):
                seq += lines[i]
                i += 1
            i += 1
            while len(qual) < len(seq) and i < len(lines):
                qual += lines[i]
                i += 1
            assert len(qual) == len(seq)
            out.append((h, seq, qual))
    return out


def test_merge_mate_pairs_random_mixes(tmp_path):
    """strict records (the reader's copy-free path), multi-line records, FASTA records, separator lines with text and
    quality lines that start with '@' or '+', in random order and with or without a final newline: the tool's output
    equals the restated reader's, record by record, for buffers of both streams that end anywhere"""
    import subprocess
    merge = os.path.join(ROOT, "quorum_amd", "bin", "merge_mate_pairs")
    rng = np.random.default_rng(8)

    def rand_file(n):
        parts = []
        fasta_only = rng.random() < 0.2
        for i in range(n):
            L = int(rng.integers(1, 300))
            seq = bytes(rng.choice(list(b"ACGTN"), size=L).tolist())
            kind = rng.random()
            if fasta_only:
                w = int(rng.integers(1, L + 1))
                parts.append(b">f%d\n" % i + b"\n".join(seq[j:j + w] for j in range(0, L, w)) + b"\n")
                continue
            qual = bytes(rng.choice(list(b"@+I#5"), size=L).tolist())
            plus = b"+" + (b"x%d" % i if rng.random() < 0.3 else b"")
            if kind < 0.8:
                parts.append(b"@r%d some text\n%s\n%s\n%s\n" % (i, seq, plus, qual))
            else:  # multi-line: the quality pieces must not start a line with '@' before the count is reached
                w = int(rng.integers(1, L + 1))
                sl = b"\n".join(seq[j:j + w] for j in range(0, L, w))
                ql = b"\n".join(qual[j:j + w] for j in range(0, L, w))
                parts.append(b"@m%d\n%s\n%s\n%s\n" % (i, sl, plus, ql))
        data = b"".join(parts)
        return data[:-1] if rng.random() < 0.5 else data

    for rep in range(6):
        n = int(rng.integers(1, 400))
        d1, d2 = rand_file(n), rand_file(n)
        (tmp_path / "x1").write_bytes(d1)
        (tmp_path / "x2").write_bytes(d2)
        r = subprocess.run([merge, str(tmp_path / "x1"), str(tmp_path / "x2")], capture_output=True, timeout=60)
        assert r.returncode == 0, r.stderr
        want = b""
        for a, b in zip(_model_parse(d1), _model_parse(d2)):
            for h, s, q in (a, b):
                want += b"@" + h + b"\n" + s + b"\n+\n" + (q if q is not None else b"*" * len(s)) + b"\n"
        assert r.stdout == want, rep
    # a large strict input crosses the 4 MiB buffer many times
    big = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, b"ACGT" * 25, b"I" * 100) for i in range(60000))
    (tmp_path / "b1").write_bytes(big)
    (tmp_path / "b2").write_bytes(big[:-1])
    r = subprocess.run([merge, str(tmp_path / "b1"), str(tmp_path / "b2")], capture_output=True, timeout=60)
    recs = big.split(b"@")[1:]
    assert r.returncode == 0 and r.stdout == b"".join(b"@" + x + b"@" + x for x in recs)
