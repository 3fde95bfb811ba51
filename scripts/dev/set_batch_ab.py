"""flowgnn_set_batch alone (one engine, 2^18 molhiv graphs = 536 MB of int32 arrays): plain copies against option h2d_pack = 1..16 host threads."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowgnn_amd import Engine, graphpack as gp, weights
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
b = gp.synth_molhiv_batch(g, seed=1234)
w = weights.synth_gin_weights(seed=7)
mb = 4 * (b.node_feature.size + b.edge_list.size + b.edge_attr.size) / 1e6
for pack in (0, 1, 2, 4, 8, 16, 32):
    e = Engine("GIN", device=0, options={"h2d_pack": pack})
    e.set_weights(w)
    e.set_batch(b)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); e.set_batch(b); ts.append(time.perf_counter() - t0)
    print(f"h2d_pack={pack:2d}: set_batch {min(ts)*1e3:.2f} ms  ({mb / min(ts) / 1e3:.1f} GB/s of int32 arrays)", flush=True)
    e.close()
