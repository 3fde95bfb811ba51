"""Per-phase clocks of gcn_resident_kernel's workgroup 0 (development build: bash scripts/dev/devlib.sh dev; gcn_ablate 64 prints them).
usage: gcn_stamps.py [graphs]   -- run on the GPU box with scripts/dev/_dev.so built"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import flowgnn_amd._lib as L
L.LIB_PATH = os.path.join(ROOT, "scripts", "dev", "_dev.so")
from flowgnn_amd import Engine, graphpack as gp, weights
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
b = gp.synth_molhiv_batch(g, seed=1234)
w = weights.synth_gcn_weights(7)
e = Engine("GCN", 0, options={"gcn_ablate": 0})
e.set_weights(w); e.set_batch(b)
for _ in range(5): e.run()
e.sync(); e.close()
e = Engine("GCN", 0, options={"gcn_ablate": 64})
e.set_weights(w); e.set_batch(b)
e.run(); e.sync(); e.close()
