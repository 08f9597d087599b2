#!/usr/bin/env python3
"""quorum -- the driver of the Quorum pipeline over the MI355X tools (src/quorum.in:17-232).

Same switches, defaults, messages and file names as the reference's Perl driver:
quorum_create_database writes <prefix>_mer_database.jf (:154-158), then quorum_error_correct_reads
writes <prefix>.fa / <prefix>.log (:160-171) or, with -P, merge_mate_pairs | quorum_error_correct_reads
-d /dev/fd/0 2> <prefix>.log | split_mate_pairs writes <prefix>_1.fa / <prefix>_2.fa (:172-231).
The quality offset is detected from the first 1000 reads when -q is not given (:129-152,
qh_detect_min_qual_char). -t is passed on and ignored by the tools (the GPU supplies the parallelism)."""
import ctypes
import getopt
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "bin")
CDB = os.path.join(BIN, "quorum_create_database")
EC = os.path.join(BIN, "quorum_error_correct_reads")
MERGE = os.path.join(BIN, "merge_mate_pairs")
SPLIT = os.path.join(BIN, "split_mate_pairs")

USAGE = """%s [options] .fastq [.fastq]+

Run the quorum error corrector on the given fastq file. If the --paired-files
switch is given, quorum expect an even number of files on the command line,
each pair files containing pair end reads. The output will be two files
(<prefix>_1.fa and <prefix>_2.fa) containing error corrected pair end reads.

Options:
 -s, --size              Mer database size (default 200M)
 -t, --threads           Number of threads (default number of cpus)
 -p, --prefix            Output prefix (default quorum_corrected)
 -k, --kmer-len          Kmer length (default 24)
 -q, --min-q-char        Minimum quality char. Usually 33 or 64 (autodetect)
 -m, --min-quality       Minimum above -q for high quality base (5)
 -w, --window            Window size for trimming
 -e, --error             Maximum number of errors in a window
     --min-count         Minimum count for a k-mer to be good
     --skip              Number of bases to skip to find anchor kmer
     --anchor            Numer of good kmer in a row for anchor
     --anchor-count      Minimum count for an anchor kmer
     --contaminant       Contaminant sequences
     --trim-contaminant  Trim sequences with contaminant mers
 -d, --no-discard        Do not discard reads, output a single N (false)
 -P, --paired-files      Preserve mate pairs in two files
     --homo-trim         Trim homo-polymer on 3' end
     --debug             Display debugging information
     --version           Display version
 -h, --help              This message
"""

SWITCHES = ["min-count", "skip", "good", "anchor-count", "window", "error", "contaminant", "homo-trim"]  # :27
FLAGS = ["trim-contaminant", "no-discard"]                                                                # :28


def detect_min_q_char(path):
    lib = ctypes.CDLL(os.path.join(HERE, "csrc", "libquorum_hip.so"))
    lib.qh_last_error.restype = ctypes.c_char_p
    import itertools
    with open(path, "rb") as f:
        head = b"".join(itertools.islice(f, 4000))  # header lines 1..4000 = the first 1000 records (quorum.in:133-144)
    c = ctypes.c_int()
    if lib.qh_detect_min_qual_char(head, ctypes.c_size_t(len(head)), ctypes.byref(c)) != 0:
        sys.stderr.write(lib.qh_last_error().decode() + "\n")
        sys.exit(1)
    return c.value


def main(argv):
    me = argv[0]
    try:
        got, files = getopt.gnu_getopt(argv[1:], "s:t:p:k:q:m:w:e:dPh", [
            "size=", "threads=", "prefix=", "klen=", "kmer-len=", "min-q-char=", "min-quality=", "window=", "error=",
            "min-count=", "skip=", "anchor=", "anchor-count=", "contaminant=", "trim-contaminant", "contaminant-trim",
            "no-discard", "paired-files", "homo-trim=", "debug", "version", "help"])
    except getopt.GetoptError as e:
        sys.stderr.write("%s\n%s" % (e, USAGE % me))
        return 1
    jf_size, prefix, klen, min_q_char, min_quality, threads = "200M", "quorum_corrected", 24, None, 5, None
    opts, paired, debug = {}, False, False
    for o, v in got:
        if o in ("-s", "--size"): jf_size = v
        elif o in ("-t", "--threads"): threads = int(v)
        elif o in ("-p", "--prefix"): prefix = v
        elif o in ("-k", "--klen", "--kmer-len"): klen = int(v)
        elif o in ("-q", "--min-q-char"): min_q_char = int(v)
        elif o in ("-m", "--min-quality"): min_quality = int(v)
        elif o in ("-w", "--window"): opts["window"] = v
        elif o in ("-e", "--error"): opts["error"] = v
        elif o == "--min-count": opts["min-count"] = v
        elif o == "--skip": opts["skip"] = v
        elif o == "--anchor": opts["good"] = v
        elif o == "--anchor-count": opts["anchor-count"] = v
        elif o == "--contaminant": opts["contaminant"] = v
        elif o in ("--trim-contaminant", "--contaminant-trim"): opts["trim-contaminant"] = True
        elif o in ("-d", "--no-discard"): opts["no-discard"] = True
        elif o in ("-P", "--paired-files"): paired = True
        elif o == "--homo-trim": opts["homo-trim"] = v
        elif o == "--debug": debug = True
        elif o == "--version":
            print("quorum_hip 0.1")
            return 0
        elif o in ("-h", "--help"):
            sys.stdout.write(USAGE % me)
            return 0
    import re
    if not re.match(r"^\d+[kMGT]?$", jf_size):  # :92-95
        sys.stderr.write("Invalid size '%s'. It must be a number, maybe followed by a suffix (like k, M, G for thousand, "
                         "million and billion).\n" % jf_size)
        return 1
    if not files:  # :96-99
        sys.stderr.write("No sequence files. See %s --help.\n" % me)
        return 1
    if threads is None:
        threads = os.cpu_count() or 1  # :110-118
    if paired and len(files) % 2:
        sys.stderr.write("Must give a even number files\n")
        return 1

    def run(cmd, **kw):
        if debug:
            sys.stderr.write("+ %s\n" % " ".join(cmd))
        return subprocess.call(cmd, **kw)

    if min_q_char is None:
        min_q_char = detect_min_q_char(files[0])
    db_file = prefix + "_mer_database.jf"
    if run([CDB, "-s", jf_size, "-m", str(klen), "-t", str(threads), "-q", str(min_q_char + min_quality), "-b", "7",
            "-o", db_file] + files) != 0:
        sys.stderr.write("Creating the mer database failed. Most likely the size passed to the -s switch is too small.\n")
        return 1
    ec_cmd = [EC, "-t", str(threads)]
    if paired:
        opts["no-discard"] = True  # :161
    for s in SWITCHES:
        if s in opts:
            ec_cmd += ["--" + s, str(opts[s])]
    for f in FLAGS:
        if f in opts:
            ec_cmd += ["--" + f]
    if not paired:
        if run(ec_cmd + [db_file, "-o", prefix] + files) != 0:
            sys.stderr.write("Error correction failed\n")
            return 1
        return 0
    if debug:
        sys.stderr.write("+ %s %s | %s %s /dev/fd/0 2> %s.log | %s %s\n"
                         % (MERGE, " ".join(files), " ".join(ec_cmd), db_file, prefix, SPLIT, prefix))
    with open(prefix + ".log", "wb") as error_log:
        merge = subprocess.Popen([MERGE] + files, stdout=subprocess.PIPE)
        correct = subprocess.Popen(ec_cmd + [db_file, "/dev/fd/0"], stdin=merge.stdout, stdout=subprocess.PIPE, stderr=error_log)
        merge.stdout.close()
        split = subprocess.Popen([SPLIT, prefix], stdin=correct.stdout)
        correct.stdout.close()
        ret = 0
        for name, p in (("merge_mate_pairs", merge), ("quorum_error_correct", correct), ("split_mate_pairs", split)):
            rc = p.wait()
            if rc != 0:
                sys.stderr.write("An error was returned by %s: %d\n" % (name, rc))
                ret = 1
    return ret


if __name__ == "__main__":
    sys.exit(main(sys.argv))
