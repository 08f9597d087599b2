#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + two PMC passes of the
# N=1 bench. Output under gpurun_out/prof_$1/ ; copy the summaries into profiles/.
set -e -o pipefail
# usage: tools/profile_bench.sh <tag> [c1|c2]
TAG=${1:-r01}
WL=${2:-c2}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-checks --no-c1 --workload $WL"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
