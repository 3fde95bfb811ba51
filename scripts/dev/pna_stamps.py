"""Per-phase times of pna_layer_fused_kernel (make DEV / scripts/dev/devlib.sh; option pna_ablate bit 128): printed on stderr by the library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import flowgnn_amd._lib as L
L.LIB_PATH = os.environ.get("FLOWGNN_LIB", os.path.join(ROOT, "scripts", "dev", "_dev.so"))
from flowgnn_amd import Engine, graphpack as gp, weights
b = gp.synth_hep10k_batch(1 << 15, seed=1234, with_eigen=False)
w = weights.synth_pna_weights(seed=7)
for ab in [int(x) for x in (sys.argv[1:] or ["128"])]:
    e = Engine("PNA", 0, options={"pna_ablate": ab})
    e.set_weights(w); e.set_batch(b)
    for _ in range(3): e.run()
    e.sync(); e.profile_enable(True)
    for _ in range(4): e.run()
    e.sync()
    print("ablate", ab, {a: round(v["total_ms"] / max(v["launches"], 1), 4) for a, v in e.profile_read().items() if "fused" in a}, flush=True)
    e.close()
