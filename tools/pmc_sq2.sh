#!/bin/bash
# instruction mix of the build kernels (one pass, 8 SQ counters)
set -e -o pipefail
OUT=$PWD/gpurun_out/pmc_sq2_$1
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --mode partition > $OUT/run.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/*/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].split("(")[0][:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "partition" in k or "window" in k or "index" in k:
        print(k, {c: round(sum(x)/len(x)/1e6,2) for c,x in v.items()})
PY
