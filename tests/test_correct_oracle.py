"""CPU tests of the corrector's oracle (oracle/error_correct_oracle.c).

PARITY UNPINNED: the reference has no test or fixture for error_correct_reads. What pins the
oracle here: (a) the output format example the reference's README holds (README.md:58-88),
(b) cases worked by hand from src/error_correct_reads.cc:246-341,386-565,609-643 and
src/err_log.hpp:55-135 -- each expected line below is derived in its comment, and the
preconditions the derivation assumes (which k-mers exist) are asserted with plain Python sets."""
import numpy as np
import pytest

import pyoracle as po
from helpers import mer_from_str, revcomp

K = 6
S = "TTACTTGCTGTGTCCACCCCATCGGACTGGC"   # 31 bases, every canonical 6-mer distinct (asserted below)
Q = "I" * len(S)


def canon(s):
    m = mer_from_str(s)
    return min(m, revcomp(m, len(s)))


def kmers(s, k=K):
    return [canon(s[i:i + k]) for i in range(len(s) - k + 1)]


def fastq(*recs):
    return "".join("@%s\n%s\n+\n%s\n" % (n, s, q) for n, s, q in recs).encode()


@pytest.fixture(scope="module")
def db():
    """five error-free high-quality copies of S: every 6-mer of S is (5, hq)"""
    d = po.OracleDB(1 << 12, K, 7)
    d.add_fastq(fastq(*[("c%d" % i, S, Q) for i in range(5)]), 38)
    return d


def correct(db, recs, **kw):
    fa, lg, n = db.correct_fastq(fastq(*recs), po.ec_params(kw.pop("cutoff", 4), **kw))
    assert n == len(recs)
    return fa.decode(), lg.decode()


def test_preconditions(db):
    ks = kmers(S)
    assert len(set(ks)) == len(ks)
    present = set(ks)
    # the derivations assume: at every position of S exactly one of the four alternatives of the last
    # (forward) / first (backward) base exists in the database
    for i in range(len(S) - K + 1):
        w = S[i:i + K]
        assert sum(canon(w[:-1] + b) in present for b in "ACGT") == 1
        assert sum(canon(b + w[1:]) in present for b in "ACGT") == 1
    m, v = db.sorted_dump()
    assert sorted(present) == m.tolist() and set(v.tolist()) == {(5 << 1) | 1}


def test_readme_log_format():
    """README.md:71: '>1204 86:sub:T-C 91:3_trunc 62:5_trunc'"""
    fwd = po.ec_print_log(+1, [(0, 86, "T", "C"), (1, 91, None, None)])
    bwd = po.ec_print_log(-1, [(1, 62, None, None)])
    assert b">1204 " + fwd + b" " + bwd == b">1204 86:sub:T-C 91:3_trunc 62:5_trunc"


def test_clean_read_is_copied(db):
    # find_starting_mer: skip 1, mers S[1:7] and S[2:8] have count 5 >= anchor 3 -> good = 2 reached with
    # input at 8. extend forward 8..30 and backward 1, 0: always one continuation equal to the read's base
    # -> no log entry. Output line: '>' header ' ' fwd_log ' ' bwd_log (error_correct_reads.cc:334-336).
    fa, lg = correct(db, [("r1", S, Q)])
    assert fa == ">r1  \n" + S + "\n" and lg == ""


def test_single_substitution(db):
    # position 20 changed (A -> C): at pos 20 the only alternative in the database is the genome's base
    # -> count == 1, ucode != ori_code -> log_substitution: "20:sub:C-A", the mer is repaired and the
    # following mers are taken from the repaired mer
    bad = S[:20] + "C" + S[21:]
    assert S[20] == "A"
    fa, lg = correct(db, [("r2", bad, Q)])
    assert fa == ">r2 20:sub:C-A \n" + S + "\n" and lg == ""


def test_n_is_substituted(db):
    # N at 15: shifted in as code 0, from = -1 prints as 'N' (error_correct_reads.cc:372)
    bad = S[:15] + "N" + S[16:]
    fa, lg = correct(db, [("r3", bad, Q)])
    assert fa == ">r3 15:sub:N-%s \n" % S[15] + S + "\n"


def test_3prime_truncation(db):
    # four foreign bases after S: at position 31 none of the four alternatives exists -> count == 0 ->
    # log.truncation(cpos = 31); the output stops before it
    tail = "ACAC"
    for b in "ACGT":
        assert canon(S[-5:] + b) not in set(kmers(S))
    fa, lg = correct(db, [("r4", S + tail, "I" * (len(S) + 4))])
    assert fa == ">r4 31:3_trunc \n" + S + "\n"


def test_5prime_truncation(db):
    # four foreign bases before S (S starts at 4). find_starting_mer: the first mers with count >= 3 are
    # S[0:6] (read 4..9) and S[1:7] -> anchor found with input at 11. Backward from 11 - 6 - 1 = 4: the mer
    # read[4:10] = S[0:6] exists -> base 4 kept; at 3 no alternative exists -> backward_log::truncation(3)
    # logs 3 + 1 (error_correct_reads.hpp:170-172): "4:5_trunc"; the output starts at 4
    head = "GTGT"
    read = head + S
    for b in "ACGT":
        assert canon(b + S[:5]) not in set(kmers(S))
    assert not (set(kmers(read[:9])) & set(kmers(S)))
    fa, lg = correct(db, [("r5", read, "I" * len(read))])
    assert fa == ">r5  4:5_trunc\n" + S + "\n"


def test_no_anchor_is_skipped(db):
    junk = "ACACACACACACACACACAC"
    assert not (set(kmers(junk)) & set(kmers(S)))
    fa, lg = correct(db, [("r6", junk, "I" * len(junk))])
    assert fa == "" and lg == "Skipped r6: No high quality mer\n"
    fa, lg = correct(db, [("r6", junk, "I" * len(junk))], no_discard=True)   # -d: a single N (:274-275)
    assert fa == ">r6\nN\n" and lg == "Skipped r6: No high quality mer\n"
    # shorter than skip + k: the loops never run
    fa, lg = correct(db, [("r7", S[:5], "IIIII")])
    assert lg == "Skipped r7: No high quality mer\n"


def test_error_window_truncates(db):
    # substitutions at 10, 13, 16, 19 with window 10, error 3: the fourth entry makes
    # size - lwin - 1 = 3 >= error (err_log.hpp:94) -> remove_last_window: diff = 19 - 10 = 9, the log is
    # emptied, out moves back 9 bases and truncation(19 - 9 = 10) is logged (error_correct_reads.cc:372-377)
    bad = list(S)
    for p in (10, 13, 16, 19):
        bad[p] = "ACGT"[("ACGT".index(S[p]) + 1) & 3]
    bad = "".join(bad)
    fa, lg = correct(db, [("r8", bad, Q)])
    assert fa == ">r8 10:3_trunc \n" + S[:10] + "\n"
    # with one more error allowed the same read is fully repaired
    fa, lg = correct(db, [("r8", bad, Q)], error=4)
    subs = " ".join("%d:sub:%s-%s" % (p, bad[p], S[p]) for p in (10, 13, 16, 19))
    assert fa == ">r8 " + subs + " \n" + S + "\n"
    # a window of 5 only ever holds two of them
    fa, lg = correct(db, [("r8", bad, Q)], window=5)
    assert fa == ">r8 " + subs + " \n" + S + "\n"


def test_backward_substitution_and_reverse_strand(db):
    # an error at position 3 lies before the anchor (the anchor search skips mers containing it: they are
    # absent): it is repaired by the backward extension and logged in the backward log
    bad = S[:3] + ("A" if S[3] != "A" else "C") + S[4:]
    fa, lg = correct(db, [("r9", bad, Q)])
    assert fa == ">r9  3:sub:%s-%s\n" % (bad[3], S[3]) + S + "\n"
    # the reverse complement of S corrects the same way (canonical lookups)
    rc = S[::-1].translate(str.maketrans("ACGT", "TGCA"))
    fa, lg = correct(db, [("r10", rc, Q)])
    assert fa == ">r10  \n" + rc + "\n"


def test_homo_trim(db):
    # homo_trim (error_correct_reads.cc:567-597) scans from the 3' end: +1 for a base equal to its
    # successor, -1 otherwise. S ends ...TGGC: 29 'G' != 'C' -> -1, 28 'G' == 'G' -> 0, 27 'T' -> -1, ...;
    # the maximum 0 is first reached at 28. With --homo-trim 0 the read is cut at 28, both logs lose their
    # entries at or after 28 and "28:3_trunc" is appended to the forward log
    assert S[27:] == "TGGC"
    fa, lg = correct(db, [("r11", S, Q)], homo_trim=0)
    assert fa == ">r11 28:3_trunc \n" + S[:28] + "\n"
    fa, lg = correct(db, [("r11", S, Q)], homo_trim=1)     # maximum score 0 < 1: untouched
    assert fa == ">r11  \n" + S + "\n"


def test_multiline_and_parse_errors(db):
    ml = ("@m1 x\n" + S[:10] + "\n" + S[10:] + "\n+\n" + Q[:7] + "\n" + Q[7:] + "\n").encode()
    fa, lg, n = db.correct_fastq(ml, po.ec_params(4))
    assert n == 1 and fa.decode() == ">m1 x  \n" + S + "\n"
    with pytest.raises(ValueError, match="wrong number of quals"):
        db.correct_fastq(b"@r\nACGT\n+\nIII\n@s\nAC\n+\nII\n", po.ec_params(4))


def test_statistical_sanity():
    """40x coverage, 1 % errors: nearly every read comes back as an exact substring of the genome"""
    from quorum_amd import synth
    g = synth.make_genome(20000, seed=1)
    fq = synth.make_fastq(g, 8000, 100, err=0.01, seed=3).tobytes()
    d = po.OracleDB(1 << 20, 24, 7)
    d.add_fastq(fq, 38)
    cut = d.poisson_cutoff(0.01 / 3, 1e-6 / 0.01)
    fa, lg, n = d.correct_fastq(fq, po.ec_params(cut))
    gs = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[g])
    grc = gs.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]
    seqs = fa.split(b"\n")[1::2]
    ok = sum(1 for s in seqs if s and (s in gs or s in grc))
    assert n == 8000 and ok > 0.97 * len([s for s in seqs if s])
    d.close()
