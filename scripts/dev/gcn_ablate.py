"""per-phase timing of gcn_resident_kernel / gat_resident_kernel with a `make DEV=1` build (FLOWGNN_LIB=...): option <model>_ablate bits
GCN: 1 no in-edge walk, 2 no dense MFMAs, 4 rows in natural order, 8 no table/epilogue blob DMA, 16 no W DMA.  usage: gcn_ablate.py [GCN|GAT]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import flowgnn_amd._lib as _L
if os.environ.get("FLOWGNN_LIB"):
    _L.LIB_PATH = os.path.abspath(os.environ["FLOWGNN_LIB"])
import bench
from flowgnn_amd import Engine, weights
model = sys.argv[1] if len(sys.argv) > 1 else "GCN"
b = bench.make_batch(bench.MODELS[model]["dataset"], 1 << 18, 1234)
w = weights.SYNTH[model](seed=7)
for ab in [0, 1, 2, 3, 4, 8, 16, 27]:
    e = Engine(model, 0, options={model.lower() + "_ablate": ab})
    e.set_weights(w); e.set_batch(b)
    for _ in range(3): e.run()
    e.sync(); e.profile_enable(True)
    for _ in range(8): e.run()
    e.sync()
    k = {a: round(v["total_ms"] / max(v["launches"], 1), 4) for a, v in e.profile_read().items() if "resident" in a}
    print(model, "ablate", ab, k, flush=True)
    e.close()
