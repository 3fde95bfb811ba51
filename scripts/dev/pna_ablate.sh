#!/bin/bash
# per-phase timing of the fused PNA / DGN layer kernels: FLOWGNN_<M>_ABLATE bit 0 = no gather loop, bit 1 = no K-steps
M=${1:-PNA}
for a in ${ABLATES:-0 1 2 3}; do
  export FLOWGNN_${M}_ABLATE=$a
  python bench.py --model $M --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read())
print('ablate', os.environ['FLOWGNN_${M}_ABLATE'], {k: round(v,3) for k,v in d['kernel_avg_ms'].items() if 'fused' in k or 'resident' in k}, d['config']['nodes_rank0'])"
done
