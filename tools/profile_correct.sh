#!/bin/bash
# Runs on the GPU box (through gpurun): the corrector benchmark (JSON line), its kernel-trace stats
# and three PMC passes. Output under gpurun_out/prof_ec_$1/ ; tools/summarize_correct.py copies the
# summaries into profiles/.
set -e -o pipefail
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_ec_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python3 tools/correct_bench.py --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
ARGS="tools/correct_bench.py --steps 3 --warmup 1 --cpu-sample 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
ARGS="tools/correct_bench.py --steps 1 --warmup 1 --cpu-sample 0"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
cat $OUT/bench.json
