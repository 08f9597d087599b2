"""qh_fastq_4line_prefix: the cut between whole 4-line records and the chunk's tail, found from the chunk's last lines
(host code, no GPU). Checked against the obvious rule -- count the newlines from the start -- on every prefix of
strict FASTQ texts whose quality lines start with '@' and '+' as often as chance allows."""
import ctypes as C

import numpy as np

from quorum_amd._lib import lib


def by_counting(buf, final):
    lines, pos, cut = 0, 0, 0
    while pos < len(buf):
        e = buf.find(b"\n", pos)
        if e < 0:
            if final and ((lines + 1) & 3) == 0:
                cut = len(buf)
            break
        pos = e + 1
        lines += 1
        if lines & 3 == 0:
            cut = pos
    return cut


def make_text(rng, n, qual_alphabet):
    recs = []
    for i in range(n):
        L = int(rng.integers(1, 40))
        seq = bytes(rng.choice(list(b"ACGTN"), size=L).tolist())
        qual = bytes(rng.choice(list(qual_alphabet), size=L).tolist())
        plus = b"+" + (b"r%d" % i if rng.random() < 0.3 else b"")
        recs.append(b"@r%d\n%s\n%s\n%s\n" % (i, seq, plus, qual))
    return b"".join(recs)


def test_cut_from_the_tail_equals_counting():
    rng = np.random.default_rng(3)
    for alphabet in (b"@+I#", b"@@@+", b"IJK5"):
        text = make_text(rng, 60, alphabet)
        for cut in range(0, len(text) + 1):
            part = text[:cut]
            for final in (0, 1):
                got = lib.qh_fastq_4line_prefix(part, len(part), final)
                assert got == by_counting(part, final), (alphabet, cut, final)
    # a last record without its final newline, and blank lines after the last record
    text = make_text(rng, 10, b"@+I#")
    for t in (text[:-1], text + b"\n", text + b"\n\n"):
        for final in (0, 1):
            assert lib.qh_fastq_4line_prefix(t, len(t), final) == by_counting(t, final)


def test_cut_of_long_lines_falls_back_to_counting():
    """lines longer than the look-back are still cut right (the count from the start takes over)"""
    seq = b"ACGT" * 5000
    text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, seq, b"I" * len(seq)) for i in range(5))
    for cut in (len(text), len(text) - 1, len(text) - len(seq) - 3, 2 * len(seq) + 30):
        for final in (0, 1):
            assert lib.qh_fastq_4line_prefix(text[:cut], cut, final) == by_counting(text[:cut], final)
