#!/bin/bash
# which node embeddings differ between two GIN layer-kernel variants (GPU box)
G=${1:-4113}
cat > /tmp/run2.py <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from flowgnn_amd import Engine, graphpack as gp, weights
w = weights.synth_gin_weights(seed=3)
b = gp.synth_molhiv_batch(int(sys.argv[1]), seed=1234)
e = Engine("GIN", device=0); e.set_weights(w)
e.forward(b)
np.save(sys.argv[2], e.final_h())
rp, src, eid, od = e.csr()
np.save('/tmp/rp.npy', rp); np.save('/tmp/src.npy', src)
PY
FLOWGNN_GIN_SPLIT_NT=1 python /tmp/run2.py $G /tmp/ha.npy
FLOWGNN_GIN_SPLIT_NT=3 python /tmp/run2.py $G /tmp/hb.npy
python - <<'PY'
import numpy as np
a, b = np.load('/tmp/ha.npy'), np.load('/tmp/hb.npy')
rp, src = np.load('/tmp/rp.npy'), np.load('/tmp/src.npy')
d = np.abs(a - b).max(axis=1)
bad = np.nonzero(d > 1e-6)[0]
print("bad nodes", len(bad), "of", len(d))
for v in bad[:60]:
    t = v // 192
    print(v, "tile", t, "local", v % 192, "wave", (v % 192) // 16, "deg", rp[v+1]-rp[v], "e_rel", rp[v]-rp[t*192], "maxerr %.3g" % d[v], "src", src[rp[v]:rp[v+1]] - t*192)
PY
