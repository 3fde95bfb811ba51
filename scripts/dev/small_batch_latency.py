"""scripts/dev/small_batch_latency.py [graphs] (GPU box): ms per flowgnn_run of a small resident batch, plain launches vs
hipGraph replay (FLOWGNN_HIPGRAPH=0 / 1), profiler off."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from flowgnn_amd import Engine, graphpack as gp, weights

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4113
for model, mk in (("GIN", gp.synth_molhiv_batch), ("GCN", gp.synth_molpcba_batch), ("GAT", gp.synth_molhiv_batch)):
    b = mk(G, seed=1234)
    w = getattr(weights, f"synth_{model.lower()}_weights")(seed=7)
    res = {}
    for mode in ("0", "1"):
        os.environ["FLOWGNN_HIPGRAPH"] = mode
        e = Engine(model, device=0)
        e.set_weights(w); e.set_batch(b)
        for _ in range(5):
            e.run()
        e.sync()
        t0 = time.perf_counter()
        n = 200
        for _ in range(n):
            e.run()
        e.sync()
        res[mode] = (time.perf_counter() - t0) / n * 1e3
        out = e.results()
        e.close()
    print(f"{model} G={G}: plain {res['0']:.4f} ms  graph {res['1']:.4f} ms  -> {G / res['1'] / 1e3:.2f} M graphs/s (plain {G / res['0'] / 1e3:.2f})")
