import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowgnn_amd import Engine, compute_graphs, graphpack as gp, weights
w = weights.synth_gin_weights(7)
b = gp.synth_molhiv_batch(1 << 18, seed=1234)
n_prior = int(sys.argv[1]) if len(sys.argv) > 1 else 1
use = len(sys.argv) > 2
for _ in range(n_prior):
    e = Engine("GIN", 0); e.set_weights(w)
    if use:
        e.set_batch(b); e.run(); e.sync()
    e.close()
ts = []
for _ in range(8):
    t0 = time.perf_counter(); compute_graphs("GIN", b, [w]); ts.append(1e3 * (time.perf_counter() - t0))
print(n_prior, use, " ".join(f"{t:.1f}" for t in ts))
