import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from flowgnn_amd import Engine, graphpack as gp, weights
b = gp.concat_batches([gp.synth_molhiv_batch(300, seed=51), gp.synth_hep10k_batch(3, seed=53, with_eigen=False), gp.synth_molhiv_batch(50, seed=54)])
w = weights.synth_gin_weights(seed=7)
for opts in ({"gin_tile_build": 1}, {"gin_tile_build": 0}, {"gin_tile_build": 0, "gin_resident": 0}):
    opts = dict(opts, gin_resident_min_fill=0)
    e = Engine("GIN", 0, options=opts)
    e.set_weights(w)
    full = e.forward(b)
    for lo, hi in ((40, 300), (0, 300), (40, 353), (1, 353)):
        part = e.forward(b.slice(lo, hi))
        bad = np.nonzero(part != full[lo:hi])[0]
        print(opts, (lo, hi), "differ:", len(bad), (bad[:8] + lo).tolist(), [int(b.nums_of_nodes[i + lo]) for i in bad[:8]], flush=True)
    e.close()
