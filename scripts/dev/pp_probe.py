"""gin_pp_kernel (gin_pingpong=1) vs the eight-wave resident kernel (=0): same bits? how fast?  usage: pp_probe.py [graphs ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from flowgnn_amd import Engine, graphpack as gp, weights
w = weights.synth_gin_weights(7)
for g in [int(a) for a in sys.argv[1:]] or [40, 300, 5000]:
    b = gp.synth_molhiv_batch(g, seed=1234)
    outs = {}
    for pp in (2, 1, 0):
        e = Engine("GIN", 0, options={"gin_pingpong": pp, "gin_tile_build": 0, "gin_resident_min_fill": 0})
        e.set_weights(w)
        e.set_batch(b)
        e.run(); e.sync()
        outs[pp] = e.results()
        e.profile_enable(True)
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            e.run()
        e.sync()
        dt = (time.perf_counter() - t0) / n * 1e3
        k = {a: round(v["total_ms"] / max(v["launches"], 1), 4) for a, v in e.profile_read().items()}
        print(f"graphs={g} pingpong={pp}: {dt:.4f} ms/step {k}", flush=True)
        e.close()
    for pp in (2, 1):
        d = np.abs(outs[pp] - outs[0])
        print(f"   pingpong={pp} bit-identical: {np.array_equal(outs[pp], outs[0])}  max|diff| {d.max():.3e}  finite {np.isfinite(outs[pp]).all()}  first {outs[pp][:3]} {outs[0][:3]}", flush=True)
