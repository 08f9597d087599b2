"""Writes tests/golden/known_answers.json.

The reference ships no golden vectors for this path (its only test draws unseeded
random sequences, unit_tests/test_mer_database.cc:13-19). These cases are
HAND-DERIVED from src/create_database.cc:64-92 and src/mer_database.hpp:102-111
(worked in SURVEY.md Appendix A); the expected tables below were typed by hand, not
produced by running any implementation. Run this script to regenerate the JSON.
"""
import json
import os

I, LQ = "I", "#"  # 'I' = 73 >= 38 (HQ), '#' = 35 < 38 (LQ)

cases = [
    # A1: N reset, quality dip, LQ->HQ reset, revcomp merge, palindromes
    dict(name="A1_n_reset_quality_dip", k=4, bits=7, min_qual=38,
         fastq="@r1\nACGTACNGTACG\n+\nIIII#IIIIIII\n",
         expected={"ACGT": [1, 1], "CGTA": [1, 1], "GTAC": [1, 1]}),
    # A2: saturation at 2^bits - 1
    dict(name="A2_saturate_b1", k=4, bits=1, min_qual=38, fastq="@r\nAAAAAAAA\n+\nIIIIIIII\n",
         expected={"AAAA": [1, 1]}),
    dict(name="A2_saturate_b2", k=4, bits=2, min_qual=38, fastq="@r\nAAAAAAAA\n+\nIIIIIIII\n",
         expected={"AAAA": [3, 1]}),
    dict(name="A2_b7_five_hq", k=4, bits=7, min_qual=38, fastq="@r\nAAAAAAAA\n+\nIIIIIIII\n",
         expected={"AAAA": [5, 1]}),
    dict(name="A2_b7_five_lq", k=4, bits=7, min_qual=38, fastq="@r\nAAAAAAAA\n+\n########\n",
         expected={"AAAA": [5, 0]}),
    dict(name="A2_lq_saturates_b2", k=4, bits=2, min_qual=38, fastq="@r\nAAAAAAAA\n+\n########\n",
         expected={"AAAA": [3, 0]}),
    # A3-style two-record interactions (canonical of ACGTA is ACGTA: revcomp TACGT is larger)
    dict(name="A3_lq_then_hq", k=5, bits=7, min_qual=38,
         fastq="@a\nACGTA\n+\n#####\n@b\nACGTA\n+\nIIIII\n", expected={"ACGTA": [1, 1]}),
    dict(name="A3_hq_then_lq", k=5, bits=7, min_qual=38,
         fastq="@a\nACGTA\n+\nIIIII\n@b\nACGTA\n+\n#####\n", expected={"ACGTA": [1, 1]}),
    dict(name="A3_lq_twice", k=5, bits=7, min_qual=38,
         fastq="@a\nACGTA\n+\n#####\n@b\nACGTA\n+\n#####\n", expected={"ACGTA": [2, 0]}),
    dict(name="A3_hq_twice", k=5, bits=7, min_qual=38,
         fastq="@a\nACGTA\n+\nIIIII\n@b\nACGTA\n+\nIIIII\n", expected={"ACGTA": [2, 1]}),
    dict(name="A3_lq_lq_hq_lq", k=5, bits=7, min_qual=38,
         fastq="@a\nACGTA\n+\n#####\n@b\nACGTA\n+\n#####\n@c\nACGTA\n+\nIIIII\n@d\nACGTA\n+\n#####\n",
         expected={"ACGTA": [1, 1]}),
    dict(name="A3_revcomp_merges", k=5, bits=7, min_qual=38,
         fastq="@a\nACGTA\n+\nIIIII\n@b\nTACGT\n+\nIIIII\n", expected={"ACGTA": [2, 1]}),
    # A4: degenerate reads
    dict(name="A4_shorter_than_k", k=5, bits=7, min_qual=38, fastq="@a\nACGT\n+\nIIII\n", expected={}),
    dict(name="A4_lowercase", k=4, bits=7, min_qual=38, fastq="@a\nacgtAC\n+\nIIIIII\n",
         expected={"ACGT": [1, 1], "CGTA": [1, 1], "GTAC": [1, 1]}),
    dict(name="A4_iupac_resets", k=3, bits=7, min_qual=38, fastq="@a\nACGRACG.ACG\n+\nIIIIIIIIIII\n",
         expected={"ACG": [3, 1]}),
    # threshold is inclusive: '&' (38) >= 38; '%' (37) is low
    dict(name="threshold_inclusive", k=3, bits=7, min_qual=38, fastq="@a\nAAC\n+\n&&&\n@b\nCCA\n+\n&%&\n",
         expected={"AAC": [1, 1], "CCA": [1, 0]}),
    # the HQ run must cover the whole window: a dip k-1 bases back still spoils it
    dict(name="hq_run_needs_k_bases", k=3, bits=7, min_qual=38, fastq="@a\nAACCA\n+\n#IIII\n",
         # windows: AAC(dip inside ->LQ) ACC(HQ) CCA(HQ); canon: AAC, ACC, CCA(TGG larger)
         expected={"AAC": [1, 0], "ACC": [1, 1], "CCA": [1, 1]}),
    # quality bytes >= 0x80 are negative as (signed) char: below any threshold
    dict(name="A4_high_bit_quality", k=3, bits=7, min_qual=38, fastq_hex=(
        "@a\nAAC\n+\n".encode().hex() + "c9c9c9" + "0a"), expected={"AAC": [1, 0]}),
    # read boundaries break windows even without a separator base
    dict(name="reads_do_not_join", k=4, bits=7, min_qual=38, fastq="@a\nAAC\n+\nIII\n@b\nCAA\n+\nIII\n",
         expected={}),
    # last line without a trailing newline
    dict(name="no_trailing_newline", k=3, bits=7, min_qual=38, fastq="@a\nAAC\n+\nIII", expected={"AAC": [1, 1]}),
]

here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "known_answers.json"), "w") as f:
    json.dump(cases, f, indent=1)
print("wrote %d cases" % len(cases))
