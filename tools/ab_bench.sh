#!/bin/bash
# A/B runs of bench.py under environment switches: tools/ab_bench.sh "VAR=a" "VAR=b" ...
for e in "$@"; do
  echo "== $e"
  env $e timeout -k 10 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']/1e9,1), {k[:16]:round(v['avg_ms'],3) for k,v in d['roofline']['kernels'].items()})" || exit 1
done
