#!/bin/bash
# scripts/dev/cmp_variants.sh [graphs]  (GPU box): logits of the GIN layer-kernel variants on one batch --
# the default eight-wave split-f16 kernel, the four-wave one, and the fp32-MFMA kernel
G=${1:-4113}
cat > /tmp/run1.py <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from flowgnn_amd import Engine, graphpack as gp, weights
w = weights.synth_gin_weights(seed=3)
b = gp.synth_molhiv_batch(int(sys.argv[1]), seed=1234)
e = Engine("GIN", device=0); e.set_weights(w)
np.save(sys.argv[2], e.forward(b))
PY
FLOWGNN_GIN_SPLIT_NT=4 python /tmp/run1.py $G /tmp/a.npy
FLOWGNN_GIN_SPLIT_NT=1 python /tmp/run1.py $G /tmp/b.npy
FLOWGNN_GIN_MFMA=f32 python /tmp/run1.py $G /tmp/c.npy
python - <<'PY'
import numpy as np
a, b, c = np.load('/tmp/a.npy'), np.load('/tmp/b.npy'), np.load('/tmp/c.npy')
d = np.abs(a - b)
print("8-wave vs 4-wave split: max", d.max(), "n_diff", (d > 0).sum())
print("8-wave split vs f32: max", np.abs(a - c).max(), " 4-wave split vs f32: max", np.abs(b - c).max(), "scale", np.abs(c).max())
PY
