"""cycle breakdown of gin_pp_kernel (option gin_resident_prof).  usage: pp_prof.py [graphs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowgnn_amd import Engine, graphpack as gp, weights
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
b = gp.synth_molhiv_batch(g, seed=1234)
e = Engine("GIN", 0, options={"gin_pingpong": int(sys.argv[2]) if len(sys.argv) > 2 else 1, "gin_tile_build": 0, "gin_resident_prof": 1})
e.set_weights(weights.synth_gin_weights(7))
e.set_batch(b)
for _ in range(3):
    e.run(); e.sync()
e.close()
