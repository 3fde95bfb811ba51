#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root:  bash profiles/run_profile.sh <tag> [bench args]
# Produces gpurun_out/<tag>_{stats,fetch,write}.txt: per-kernel time (rocprofv3 --kernel-trace --stats)
# and HBM-side traffic counters, each PMC counter in its own pass (no trace domains mixed in).
set -u
TAG=${1:-prof}; shift || true
ARGS=${@:---steps 3 --warmup 1 --no-cpu-baseline}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -o kt -- python $R/bench.py $ARGS > $OUT/${TAG}_kt.log 2>&1
f=$(find $OUT/${TAG}_kt -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && python $R/profiles/summarize.py stats $f > $OUT/${TAG}_stats.txt
f=$(find $OUT/${TAG}_kt -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python $R/profiles/summarize.py stats $f > $OUT/${TAG}_trace_summary.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/${TAG}_$C -o pmc -- python $R/bench.py $ARGS > $OUT/${TAG}_$C.log 2>&1
  f=$(find $OUT/${TAG}_$C -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python $R/profiles/summarize.py pmc $f $C > $OUT/${TAG}_$C.txt
done
tail -2 $OUT/${TAG}_kt.log
rm -rf $OUT/${TAG}_kt $OUT/${TAG}_FETCH_SIZE $OUT/${TAG}_WRITE_SIZE
ls $OUT
