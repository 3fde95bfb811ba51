"""GIN step, kernel by kernel, for the one-pass front end (gin_tile_build=1) and the three-kernel one (=0); optional phase stamps.
usage: gin_paths.py [graphs] [prof]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import flowgnn_amd._lib as _L
if os.environ.get("FLOWGNN_LIB"):  # a variant build (scripts/dev/variant.sh)
    _L.LIB_PATH = os.path.abspath(os.environ["FLOWGNN_LIB"])
from flowgnn_amd import Engine, graphpack as gp, weights
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
prof = "prof" in sys.argv[2:]
model = "GIN-VN" if "vn" in sys.argv[2:] else "GIN"
b = gp.synth_molhiv_batch(g, seed=1234)
if model == "GIN-VN":
    b = gp.add_virtual_nodes(b)
w = weights.synth_gin_weights(7)
for tb in (1, 0):
    opts = {"gin_tile_build": tb}
    if prof:
        opts["gin_resident_prof"] = 1
    e = Engine(model, 0, options=opts)
    e.set_weights(w)
    e.set_batch(b)
    for _ in range(2):
        e.run()
    e.sync()
    if not prof:
        e.profile_enable(True)
    t0 = time.perf_counter()
    n = 3 if prof else 20
    for _ in range(n):
        e.run()
    e.sync()
    dt = (time.perf_counter() - t0) / n * 1e3
    k = {} if prof else {a: round(v["total_ms"] / max(v["launches"], 1), 4) for a, v in e.profile_read().items()}
    print(f"{model} tile_build={tb} graphs={g}: {dt:.4f} ms/step  {k}", flush=True)
    e.close()
