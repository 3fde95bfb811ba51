#!/bin/bash
# bash profiles/run_pmc.sh <tag> "<COUNTER1 COUNTER2 ...>" [bench args]   (GPU box, via gpurun)
# One rocprofv3 --pmc pass (counters only, no trace domains); prints per-kernel averages.
set -u
TAG=$1; CTRS=$2; shift 2
ARGS=${@:---steps 2 --warmup 1 --no-cpu-baseline}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d $OUT/${TAG}_pmc -o pmc -- python $R/bench.py $ARGS > $OUT/${TAG}_pmc.log 2>&1
f=$(find $OUT/${TAG}_pmc -name '*counter_collection.csv' | head -1)
python - "$f" > $OUT/${TAG}_pmc.txt <<'PY'
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
rows = list(csv.DictReader(open(sys.argv[1])))
print("# columns:", ",".join(rows[0].keys()) if rows else "")
for r in rows:
    a = agg[r["Kernel_Name"]][r["Counter_Name"]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
    if "Start_Timestamp" in r and "End_Timestamp" in r:
        d = agg[r["Kernel_Name"]]["(duration_ns)"]
        d[0] += 1; d[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, cs in agg.items():
    print(k[:90])
    for c, (n, t) in sorted(cs.items()):
        print(f"    {c:<34} dispatches={n:<5} avg={t/n:,.1f}")
PY
rm -rf $OUT/${TAG}_pmc
cat $OUT/${TAG}_pmc.txt
