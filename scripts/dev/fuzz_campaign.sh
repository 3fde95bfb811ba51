#!/bin/bash
# The round's differential fuzz campaign: every model x {float, option variants, entry points, fixed point}, SECONDS each, one box.
#   gpurun --timeout 2400 -- 'bash scripts/dev/fuzz_campaign.sh 70 606 > gpurun_out/rNN_fuzz_campaign.log 2>&1'
S=${1:-70}; SEED=${2:-606}
for M in GIN GIN-VN GCN GAT PNA DGN; do
  for MODE in f32 variants entry q; do
    [ $M = GIN-VN ] && [ $MODE = q ] && continue
    timeout $((S + 120)) python scripts/dev/fuzz.py $M $S $SEED $MODE 2>&1 | tail -2
  done
done
