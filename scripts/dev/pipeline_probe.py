"""flowgnn_group_compute on ONE device: wall time by engine count and ranges per engine.  usage: pipeline_probe.py [MODEL] [graphs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from flowgnn_amd import EngineGroup, graphpack as gp, weights
model = sys.argv[1] if len(sys.argv) > 1 else "GIN"
g = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
hep = model in ("PNA", "DGN")
w = getattr(weights, "synth_%s_weights" % model.lower().replace("-vn", ""))(7)
b = (gp.synth_hep10k_batch if hep else gp.synth_molhiv_batch)(g, seed=1234)
ref = None
for engines, chunks in ((1, 1), (1, 4), (2, 1), (2, 2), (2, 4), (2, 8), (3, 4)):
    grp = EngineGroup(model, [0] * engines)
    grp.set_weights(w)
    out = grp.compute(b, chunks)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); out = grp.compute(b, chunks); ts.append(time.perf_counter() - t0)
    if ref is None:
        ref = out
    print(f"{model} graphs={g} engines={engines} chunks/engine={chunks}: {min(ts) * 1e3:.2f} ms  (max|diff| vs 1x1 {np.abs(out - ref).max():.2e})", flush=True)
    grp.close()
