"""Host cost of the multi-device group's call path (persistent worker per engine) on ONE GPU listed n times.
Per graph count: one engine vs groups of 2 / 4 / 8 engines on device 0 -- wall time per step of `run` x STEPS + one sync, and the
host time of the run call alone (enqueue only).   usage: group_overhead.py [model] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from flowgnn_amd import Engine, EngineGroup, graphpack as gp, weights

model = sys.argv[1] if len(sys.argv) > 1 else "GIN"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
w = weights.SYNTH[model](seed=7)
for graphs in (4113, 1 << 15, 1 << 18):
    b = gp.synth_molhiv_batch(graphs, seed=1234)
    row = {}
    for n in (1, 2, 4, 8):
        obj = Engine(model, 0) if n == 1 else EngineGroup(model, [0] * n)
        obj.set_weights(w)
        obj.set_batch(b)
        st = steps if graphs < (1 << 18) else max(steps // 10, 10)
        for _ in range(5):
            obj.run()
        obj.sync()
        t0 = time.perf_counter()
        call = 0.0
        for _ in range(st):
            c0 = time.perf_counter()
            obj.run()
            call += time.perf_counter() - c0
        obj.sync()
        wall = (time.perf_counter() - t0) / st
        row[n] = (wall * 1e6, call / st * 1e6)
        obj.close()
    base = row[1][0]
    print(f"{model} {graphs} graphs: " + " | ".join(f"x{n}: {row[n][0]:.1f} us/step ({row[n][0] / base:.3f} of one engine), run call {row[n][1]:.1f} us" for n in row), flush=True)
