#!/bin/bash
# timing-only ablations of the partitioned build (results are wrong on purpose)
for a in 0 1 2 3 16 32 48 64 112; do
  QH_ABLATE=$a timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --mode partition --no-cpu-baseline > gpurun_out/abl_$a.log 2>&1
  python -c "
import json; d=json.loads(open('gpurun_out/abl_$a.log').read().strip().split(chr(10))[-1]); print($a, round(d['ms_per_step'],3), d['roofline']['all_kernels_avg_ms'])"
done
