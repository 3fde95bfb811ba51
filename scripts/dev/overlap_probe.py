"""Does a host -> device copy (flowgnn_set_batch of engine B) proceed while engine A's persistent kernels run?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowgnn_amd import Engine, graphpack as gp, weights
w = weights.synth_gin_weights(7)
b = gp.synth_molhiv_batch(1 << 18, seed=1234)
ea, eb = Engine("GIN", 0, options={"hipgraph": 0}), Engine("GIN", 0)
for e in (ea, eb):
    e.set_weights(w); e.set_batch(b); e.run(); e.sync()
t0 = time.perf_counter(); eb.set_batch(b); t_alone = time.perf_counter() - t0
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        ea.run()
    t1 = time.perf_counter(); eb.set_batch(b); t2 = time.perf_counter(); ea.sync(); t3 = time.perf_counter()
    print(f"queue 10 runs {1e3 * (t1 - t0):.2f} ms | set_batch under them {1e3 * (t2 - t1):.2f} ms (alone {1e3 * t_alone:.2f}) | until A is done {1e3 * (t3 - t0):.2f} ms", flush=True)
