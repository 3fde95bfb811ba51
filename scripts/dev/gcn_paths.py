"""GCN step kernel by kernel for the one-pass and the three-launch front end (option gcn_tile_build), molpcba-shaped batch.
usage: gcn_paths.py [graphs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from flowgnn_amd import Engine, graphpack as gp, weights
graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
b = gp.synth_molpcba_batch(graphs, seed=1234)
w = weights.synth_gcn_weights(seed=7)
outs = {}
for rep in range(2):
    for tb in (0, 1):
        e = Engine("GCN", 0, options={"gcn_tile_build": tb})
        e.set_weights(w); e.set_batch(b)
        for _ in range(8): e.run()
        e.sync()
        t0 = time.perf_counter()
        for _ in range(20): e.run()
        e.sync()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        e.profile_enable(True)
        for _ in range(10): e.run()
        e.sync()
        k = {a: round(v["total_ms"] / max(v["launches"], 1), 4) for a, v in e.profile_read().items()}
        outs[tb] = e.results().copy()
        print(f"gcn_tile_build={tb}: {dt:.4f} ms/step  {k}", flush=True)
        e.close()
d = np.abs(outs[0] - outs[1])
print("one-pass vs three-launch: max |d| =", float(d.max()), "max |out| =", float(np.abs(outs[0]).max()))
