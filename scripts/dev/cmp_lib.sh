#!/bin/bash
# scripts/dev/cmp_lib.sh <lib.so>: logits of that library (tile-staged kernel) vs the default .so with the per-workgroup kernel
G=${2:-4113}
cat > /tmp/run3.py <<'PY'
import os, sys, shutil, numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
from flowgnn_amd import Engine, graphpack as gp, weights
w = weights.synth_gin_weights(seed=3)
b = gp.synth_molhiv_batch(int(sys.argv[1]), seed=1234)
e = Engine("GIN", device=0); e.set_weights(w)
np.save(sys.argv[2], e.forward(b))
PY
cd $GRAFT_REPO_ROOT
FLOWGNN_GIN_SPLIT_NT=1 python /tmp/run3.py $G /tmp/a.npy
cp flowgnn_amd/libflowgnn_hip.so /tmp/orig.so
cp $1 flowgnn_amd/libflowgnn_hip.so
FLOWGNN_GIN_SPLIT_NT=3 python /tmp/run3.py $G /tmp/b.npy
cp /tmp/orig.so flowgnn_amd/libflowgnn_hip.so
python -c "
import numpy as np
a,b=np.load('/tmp/a.npy'),np.load('/tmp/b.npy'); d=np.abs(a-b); print('$1', 'max', d.max(), 'n_diff', (d>1e-6).sum())"
