"""Why was GIN-VN slower inside bench.py's `configs` pass than alone?  Same process, different orders."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from flowgnn_amd import graphpack as gp
mol = bench.make_batch("molhiv", 1 << 18, 1234)
vn = gp.add_virtual_nodes(mol)
for name, model, b in [("GIN-VN first", "GIN-VN", vn), ("GAT", "GAT", mol), ("GIN-VN after GAT", "GIN-VN", vn), ("GIN", "GIN", mol), ("GIN-VN after GIN", "GIN-VN", vn)]:
    r = bench.measure_config(model, b, 10, 2, 0)
    print(name, r["ms_per_step"], r["avg_ms"], flush=True)
