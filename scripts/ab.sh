#!/bin/bash
# usage: scripts/ab.sh "<ENV=val ...>" ["<ENV=val ...>" ...] -- prints graphs/s, ms/step and per-kernel ms for each env setting
for e in "$@"; do
  echo "== $e"
  env $e timeout 300 python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline ${BENCH_ARGS} 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_avg_ms'].items()})"
done
