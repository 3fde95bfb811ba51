"""per-phase timing of dgn_layer_mfma_kernel (needs `make DEV=1`): option dgn_ablate bits 1 no aggregation MFMAs, 2 no dense MFMAs,
4 no transposing stores, 8 no in-edge pass, 16 no h[v] loads.  usage: dgn_ablate.py [model PNA|DGN]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from flowgnn_amd import Engine, weights
model = sys.argv[1] if len(sys.argv) > 1 else "DGN"
b = bench.make_batch("hep10k", 1 << 15, 1234)
w = weights.SYNTH[model](seed=7)
for ab in ([0, 1, 2, 3, 4, 8, 16, 31] if model == "DGN" else [0, 1, 2, 3, 4, 5, 7]):
    e = Engine(model, 0, options={model.lower() + "_ablate": ab})
    e.set_weights(w); e.set_batch(b)
    for _ in range(2): e.run()
    e.sync(); e.profile_enable(True)
    for _ in range(5): e.run()
    e.sync()
    k = {a: round(v["total_ms"] / max(v["launches"], 1), 4) for a, v in e.profile_read().items() if "fused" in a}
    print(model, "ablate", ab, k, flush=True)
    e.close()
