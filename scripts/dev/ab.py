"""A/B of two builds of the library on ONE box (boxes differ by 1-2 %, so two gpurun calls cannot separate small changes).
usage: ab.py MODEL graphs libA.so libB.so [rounds] [key=value | A:key=value | B:key=value ...]
       (each measurement in its own process, A and B alternating; an A: / B: prefix gives the option to that side only, so the same
       library can be compared with itself under two option sets: ab.py PNA 32768 lib.so lib.so 3 B:pna_resident=0)
Make the B build with:  make -C flowgnn_amd/csrc LIB=../../scripts/dev/_b.so HOST=  (after `rm *.o`), *.so is git-ignored."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
import flowgnn_amd._lib as L
L.LIB_PATH = sys.argv[3]
from flowgnn_amd import Engine, graphpack as gp, weights
model, g = sys.argv[1], int(sys.argv[2])
opts = dict(kv.split("=") for kv in sys.argv[4:])
opts = {k: float(v) for k, v in opts.items()}
hep = model in ("PNA", "DGN")
b = (gp.synth_hep10k_batch if hep else gp.synth_molhiv_batch)(g, seed=1234)
if model == "GIN-VN":
    b = gp.add_virtual_nodes(b)
w = getattr(weights, "synth_%%s_weights" %% model.lower().replace("-vn", ""))(7)
e = Engine(model, 0, options=opts)
e.set_weights(w); e.set_batch(b)
for _ in range(8): e.run()
e.sync(); e.profile_enable(True)
t0 = time.perf_counter()
for _ in range(20): e.run()
e.sync()
dt = (time.perf_counter() - t0) / 20 * 1e3
k = {a: round(v["total_ms"] / max(v["launches"], 1), 4) for a, v in e.profile_read().items()}
print("%%.4f %%s" %% (dt, k))
''' % ROOT
model, g, la, lb = sys.argv[1:5]
rest = sys.argv[5:]
rounds = int(rest.pop(0)) if rest and rest[0].isdigit() else 3
for r in range(rounds):
    for name, lib in (("A", la), ("B", lb)):
        mine = [a[2:] if a[:2] == name + ":" else a for a in rest if a[:2] in (name + ":",) or a[1:2] != ":"]
        out = subprocess.run([sys.executable, "-c", CHILD, model, g, os.path.abspath(lib)] + mine, capture_output=True, text=True)
        print(name, out.stdout.strip() or out.stderr.strip()[-400:], flush=True)
