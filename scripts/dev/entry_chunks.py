"""The drop-in symbol with host arrays by pipeline setting (ranges per engine; 0 = by size, the default): usage entry_chunks.py [MODEL] [graphs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowgnn_amd import compute_graphs, entry_set_pipeline, graphpack as gp, weights
model = sys.argv[1] if len(sys.argv) > 1 else "GIN"
g = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
hep = model in ("PNA", "DGN")
b = (gp.synth_hep10k_batch if hep else gp.synth_molhiv_batch)(g, seed=1234)
w = getattr(weights, "synth_%s_weights" % model.lower())(7)
for chunks in (0, 1, 2, 3, 4, 6, 8, 0):
    entry_set_pipeline(chunks)
    compute_graphs(model, b, [w])
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); compute_graphs(model, b, [w]); ts.append(time.perf_counter() - t0)
    print(f"{model} {g} graphs, ranges per engine {chunks}: min {min(ts) * 1e3:.2f} ms, median {sorted(ts)[2] * 1e3:.2f} ms", flush=True)
