"""Phase breakdown of the graph-resident GIN kernel (FLOWGNN_GIN_RESIDENT_PROF=1 prints it per launch).
usage: resident_prof.py [graphs] [zero]   -- `zero`: all-zero weights (no operand toggling: separates power throttling from issue stalls)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["FLOWGNN_GIN_RESIDENT_PROF"] = "1"
import numpy as np
from flowgnn_amd import Engine, graphpack as gp, weights
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
b = gp.synth_molhiv_batch(g, seed=1234)
vn = "vn" in sys.argv[2:]   # GIN-VN: one virtual node per graph (a hub row of in-degree n), forced through the resident kernel
if vn:
    b = gp.add_virtual_nodes(b)
    os.environ["FLOWGNN_GIN_RESIDENT"] = "1"
e = Engine("GIN-VN" if vn else "GIN", 0)
w = weights.synth_gin_weights(7)
if "zero" in sys.argv[2:]:
    w = {k: np.zeros_like(v) for k, v in w.items()}
if "attr0" in sys.argv[2:]:   # every edge the same code: the edge-embedding reads of a ds_read group all hit one row (no bank conflicts)
    b.edge_attr[:] = 0
if "natural" in sys.argv[2:]:
    os.environ["FLOWGNN_GIN_RESIDENT_NOSORT"] = "1"
e.set_weights(w)
e.set_batch(b)
for _ in range(3):
    e.run(); e.sync()
