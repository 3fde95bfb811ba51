#!/bin/bash
# scripts/dev/kernel_times.sh [bench args]  (GPU box): rocprofv3 kernel-trace averages (ms) of one bench run, per kernel
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(f'{r["Name"][:70]:<70} {r["Calls"]:>4} {float(r["AverageNs"]) / 1e6:8.3f} ms')
PY
