"""Quick GPU sanity run: small synthetic chunk -> GPU build -> compare with the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import quorum_amd as qa
from quorum_amd import synth
import pyoracle as po

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
k, bits, thresh = 24, 7, 38
g = synth.make_genome(200000)
fq = synth.make_fastq(g, n_reads, 100, tails=True).tobytes()
t = time.time()
db = po.OracleDB(1 << 22, k, bits)
n = db.add_fastq(fq, thresh, threads=8)
print("oracle: %d instances, %d distinct, %.2fs" % (n, db.distinct(), time.time() - t), flush=True)
h = qa.HashWithQuality(1 << 22, 2 * k, bits, min_qual=thresh)
if len(sys.argv) > 2:
    h.set_mode(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else -1)
t = time.time()
h.add_fastq(fq)
st = h.done()
print("gpu:", st, "%.3fs" % (time.time() - t), flush=True)
m, v = h.dump()
o = np.argsort(m)
m, v = m[o], v[o]
om, ov = db.sorted_dump()
print("distinct equal:", m.size == om.size)
ok = m.size == om.size and (m == om).all() and (v == ov).all()
print("PARITY", "OK" if ok else "MISMATCH")
if not ok:
    print(m[:5], om[:5], v[:5], ov[:5])
    sys.exit(1)
