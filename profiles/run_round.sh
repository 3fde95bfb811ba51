#!/bin/bash
# One command that regenerates the round's evidence for every BASELINE config (GPU box, via gpurun, from the repo root):
#     bash profiles/run_round.sh r02 [models...]        (default: GIN GIN-VN GCN GAT PNA DGN)
# For each model it writes under gpurun_out/ (copy what is to be judged into profiles/):
#   <tag>_bench_<M>.json              the bench.py line (10 timed steps, CPU baseline and parity included)
#   <tag>_<M>_kernel_trace_summary.txt  rocprofv3 --kernel-trace, timed launches only (warm-up excluded, summarize.py)
#   <tag>_<M>_pmc_{FETCH,WRITE}_SIZE.txt  HBM-side traffic, one counter per pass (no trace domains mixed in)
#   <tag>_<M>_pmc_{SQ,SQ2}.txt           SQ counters per kernel (profiles/make_limits.py turns them into profiles/limits.json)
set -u
TAG=${1:-r02}; shift || true
MODELS=${@:-GIN GIN-VN GCN GAT PNA DGN}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
STEPS=10; WARM=8
cd /tmp && export TMPDIR=/tmp
for M in $MODELS; do
  python $R/bench.py --model $M --steps 20 --warmup 8 --configs off > $OUT/${TAG}_bench_$M.json 2> $OUT/${TAG}_bench_$M.err
  rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_${M}_kt -o kt -- python $R/bench.py --model $M --steps $STEPS --warmup $WARM --no-cpu-baseline --no-entry-point --configs off > $OUT/${TAG}_${M}_kt.log 2>&1
  f=$(find $OUT/${TAG}_${M}_kt -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python $R/profiles/summarize.py stats $f $STEPS $WARM > $OUT/${TAG}_${M}_kernel_trace_summary.txt
  rm -rf $OUT/${TAG}_${M}_kt
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/${TAG}_${M}_$C -o pmc -- python $R/bench.py --model $M --steps 2 --warmup 1 --no-cpu-baseline --no-entry-point --configs off > $OUT/${TAG}_${M}_$C.log 2>&1
    f=$(find $OUT/${TAG}_${M}_$C -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python $R/profiles/summarize.py pmc $f $C > $OUT/${TAG}_${M}_pmc_$C.txt
    rm -rf $OUT/${TAG}_${M}_$C
  done
  # SQ counters of the same kernels, two passes of <= 8 counters (matrix pipe / LDS / waiting; VALU / LDS instruction issue)
  for P in 1 2; do
    if [ $P = 1 ]; then CT="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES"; SUF=SQ;
    else CT="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; SUF=SQ2; fi
    rocprofv3 --pmc $CT --output-format csv -d $OUT/${TAG}_${M}_$SUF -o pmc -- python $R/bench.py --model $M --steps 2 --warmup 1 --no-cpu-baseline --no-entry-point --configs off > $OUT/${TAG}_${M}_$SUF.log 2>&1
    f=$(find $OUT/${TAG}_${M}_$SUF -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python $R/profiles/summarize.py pmcall $f > $OUT/${TAG}_${M}_pmc_$SUF.txt
    rm -rf $OUT/${TAG}_${M}_$SUF
  done
  python - $OUT/${TAG}_bench_$M.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{d['metric']}: {d['value']/1e6:.2f} M graphs/s, {d['ms_per_step']:.3f} ms/step, parity {d.get('parity')}")
PY
done
ls $OUT | grep "^${TAG}_" | head -60
