"""End-to-end timing of the quorum_create_database tool on the bench FASTQ (file in page cache):
wall clock of the whole command, and the k-mers/s that makes. Run on the GPU box."""
import argparse, json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quorum_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cli = os.path.join(ROOT, "quorum_amd", "bin", "quorum_create_database")
ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=2_000_000)
ap.add_argument("--genome", type=int, default=5_000_000)
ap.add_argument("--size", default="50M")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--no-correct", action="store_true")
ap.add_argument("--paired", action="store_true", help="also time the paired-end plumbing: merge_mate_pairs | corrector -d | split_mate_pairs")
args = ap.parse_args()
path = "/tmp/cli_bench.fastq"
g = synth.make_genome(args.genome)
fq = synth.make_fastq(g, args.reads, 100)
fq.tofile(path)
out = "/tmp/cli_bench.db"
res = []
for it in range(args.reps):
    t0 = time.time()
    r = subprocess.run([cli, "-s", args.size, "-m", "24", "-b", "7", "-q", "38", "-o", out, path],
                       capture_output=True, text=True, env=dict(os.environ, QUORUM_HIP_VERBOSE="1"))
    dt = time.time() - t0
    assert r.returncode == 0, r.stderr
    res.append(dt)
kmers = int(r.stderr.split("k-mers")[1].split()[0])
print(json.dumps({"fastq_bytes": int(fq.size), "kmers": kmers, "wall_s": res, "best_kmers_per_s": kmers / min(res),
                  "db_bytes": os.path.getsize(out), "stderr": r.stderr.strip()}))

if args.no_correct:
    sys.exit(0)
# the corrector on the same file and database (src/quorum.in:160-171)
ec = os.path.join(ROOT, "quorum_amd", "bin", "quorum_error_correct_reads")
eres = []
for it in range(args.reps):
    t0 = time.time()
    r = subprocess.run([ec, "-v", out, "-o", "/tmp/cli_bench_ec", path], capture_output=True, text=True)
    dt = time.time() - t0
    assert r.returncode == 0, r.stderr
    eres.append(dt)
print(json.dumps({"tool": "quorum_error_correct_reads", "reads": args.reads, "wall_s": eres,
                  "best_bases_per_s": args.reads * 100 / min(eres), "fa_bytes": os.path.getsize("/tmp/cli_bench_ec.fa"),
                  "stderr": r.stderr.strip().splitlines()}))

if args.paired:
    # src/quorum.in:172-231: the two halves of the reads as mates
    rb = synth.record_bytes(100)
    half = (args.reads // 2) * rb
    fq[:half].tofile("/tmp/cli_bench_1.fastq")
    fq[half:2 * half].tofile("/tmp/cli_bench_2.fastq")
    bindir = os.path.dirname(cli)
    cmd = ("%s/merge_mate_pairs /tmp/cli_bench_1.fastq /tmp/cli_bench_2.fastq | %s/quorum_error_correct_reads -d %s /dev/fd/0 "
           "2> /tmp/cli_bench_pe.log | %s/split_mate_pairs /tmp/cli_bench_pe" % (bindir, bindir, out, bindir))
    pres = []
    for it in range(args.reps):
        t0 = time.time()
        r = subprocess.run(["bash", "-o", "pipefail", "-c", cmd], capture_output=True, text=True)
        dt = time.time() - t0
        assert r.returncode == 0, r.stderr
        pres.append(dt)
    print(json.dumps({"tool": "merge_mate_pairs | quorum_error_correct_reads -d | split_mate_pairs", "reads": 2 * (args.reads // 2),
                      "wall_s": pres, "best_bases_per_s": 2 * (args.reads // 2) * 100 / min(pres),
                      "fa_bytes": [os.path.getsize("/tmp/cli_bench_pe_1.fa"), os.path.getsize("/tmp/cli_bench_pe_2.fa")]}))
