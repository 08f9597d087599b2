#!/bin/bash
# PMC passes over the corrector benchmark (one pass per counter group, as rocprofv3 cannot split them)
set -e -o pipefail
OUT=$PWD/gpurun_out/pmc_correct_$1
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="tools/correct_bench.py --steps 1 --warmup 1 --cpu-sample 0"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
python3 - <<PY
import csv,glob,collections
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        if "k_correct" in k or k.startswith("k_ec_format") or k.startswith("k_random"):
            print(k, {c: round(sum(x)/len(x)/1e6,3) for c,x in v.items()}, "(millions, avg per launch)")
PY
