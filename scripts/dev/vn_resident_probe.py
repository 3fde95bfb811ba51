"""Timing probe: GIN-VN batch through the graph-resident kernel (FLOWGNN_GIN_RESIDENT=1), kernel-level times from the engine's profiler."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["FLOWGNN_GIN_RESIDENT"] = sys.argv[1] if len(sys.argv) > 1 else "1"
from flowgnn_amd import Engine, graphpack as gp, weights
b = gp.add_virtual_nodes(gp.synth_molhiv_batch(1 << 18, seed=1234))
e = Engine("GIN-VN", 0)
e.set_weights(weights.synth_gin_weights(7))
e.set_batch(b)
for _ in range(2):
    e.run(); e.sync()
e.profile_enable(True)
for _ in range(5):
    e.run()
e.sync()
print({k: round(v["total_ms"] / max(v["launches"], 1), 3) for k, v in e.profile_read().items()})
