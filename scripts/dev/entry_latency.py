"""Wall time of the drop-in boundary with HOST buffers: flowgnn_set_batch (validation, tile packing, H2D) and the whole
<M>_compute_graphs entry call, per model and batch size.  usage: entry_latency.py [MODEL] [graphs ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from flowgnn_amd import Engine, compute_graphs, graphpack as gp, weights
model = sys.argv[1] if len(sys.argv) > 1 else "GIN"
sizes = [int(a) for a in sys.argv[2:]] or [4113, 1 << 18]
hep = model in ("PNA", "DGN")
w = getattr(weights, "synth_%s_weights" % model.lower().replace("-vn", ""))(7)
for g in sizes:
    b = (gp.synth_hep10k_batch if hep else gp.synth_molhiv_batch)(g, seed=1234)
    if model == "GIN-VN":
        b = gp.add_virtual_nodes(b)
    nbytes = b.node_feature.nbytes + b.edge_list.nbytes + (b.edge_attr.nbytes if b.edge_attr is not None else 0)
    e = Engine(model, 0)
    e.set_weights(w)
    e.set_batch(b); e.run(); e.sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); e.set_batch(b); t1 = time.perf_counter(); e.run(); e.sync(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    sb, rn = min(t[0] for t in ts) * 1e3, min(t[1] for t in ts) * 1e3
    e.close()
    compute_graphs(model, b, [w])
    tt = []
    for _ in range(3):
        t0 = time.perf_counter(); compute_graphs(model, b, [w]); tt.append(time.perf_counter() - t0)
    ent = min(tt) * 1e3
    print(f"{model} graphs={g}: host arrays {nbytes / 1e6:.1f} MB | set_batch {sb:.2f} ms ({nbytes / sb / 1e6:.1f} GB/s) | run+sync {rn:.3f} ms | "
          f"entry point {ent:.2f} ms = {g / ent / 1e3:.2f} M graphs/s", flush=True)
