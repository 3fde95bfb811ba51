"""Bisect the round-3 finding that PNA's weight-chunk LDS-DMA, requested BETWEEN a wave's two phases, made multi-engine runs differ
in the last bit from run to run (needs scripts/dev/_dev.so = scripts/dev/devlib.sh; option pna_ablate bits 8 / 16 / 32 / 64, pna.hip).
Per variant: (A) the group test's procedure (one engine vs groups {0,0} / {0,0,0}, small ragged batch) REPS times;
(B) the same batch on 1, 2 and 4 concurrent engines; (C) launch time at 2^15 graphs.   usage: pna_dma_race.py [reps]"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import flowgnn_amd._lib as L
L.LIB_PATH = os.environ.get("FLOWGNN_LIB", os.path.join(ROOT, "scripts", "dev", "_dev.so"))
from flowgnn_amd import Engine, EngineGroup, graphpack as gp, weights

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
w = weights.synth_pna_weights(seed=7)
small = gp.synth_hep10k_batch(30, seed=44, with_eigen=False)
mid = gp.synth_hep10k_batch(3000, seed=144, with_eigen=False)
big = gp.synth_hep10k_batch(1 << 15, seed=1234, with_eigen=False)


HAS_ABLATE = True
try:
    Engine("PNA", 0, options={"pna_ablate": 0}).close()
except Exception:
    HAS_ABLATE = False  # a shipped-style build (e.g. variant.sh -DPNA_DMA_MID=1): only the placement compiled in


def opts(ab):
    return {"pna_ablate": ab} if HAS_ABLATE else {}


def one(b, ab):
    e = Engine("PNA", 0, options=opts(ab))
    try:
        e.set_weights(w)
        return e.forward(b).copy()
    finally:
        e.close()


def describe(got, want):
    d = np.nonzero(got != want)[0]
    if d.size == 0:
        return None
    rel = np.abs(got[d] - want[d]) / np.maximum(np.abs(want[d]), 1e-30)
    return (int(d.size), float(rel.max()))


base = {"small": one(small, 0), "mid": one(mid, 0)}
for ab in ((0, 8, 8 + 16, 8 + 32, 8 + 64, 8 + 16 + 32 + 64) if HAS_ABLATE else (0,)):
    # variants agree with the shipped placement?  (quiet, single engine)
    q_small, q_mid = one(small, ab), one(mid, ab)
    print(f"ablate {ab}: quiet vs shipped placement: small {describe(q_small, base['small'])} mid {describe(q_mid, base['mid'])}", flush=True)
    rep_quiet = [describe(one(mid, ab), q_mid) for _ in range(3)]
    print(f"  quiet repeat (mid): {rep_quiet}", flush=True)
    # (A) group procedure
    resA = []
    for r in range(reps):
        for devs in ([0, 0], [0, 0, 0]):
            g = EngineGroup("PNA", devs, options=opts(ab))
            try:
                g.set_weights(w)
                got = g.forward(small)
                resA.append(describe(got, q_small))
                g.run()
                resA.append(describe(g.results(), q_small))
                got2 = g.forward(mid)
                resA.append(describe(got2, q_mid))
            finally:
                g.close()
    badA = [x for x in resA if x]
    print(f"  (A) groups: {len(badA)} of {len(resA)} differ; examples {badA[:5]}", flush=True)
    # (B) concurrent engines, same batch
    for n in (1, 2, 4):
        bad = []
        def worker():
            e = Engine("PNA", 0, options=opts(ab))
            try:
                e.set_weights(w); e.set_batch(mid)
                for _ in range(60):
                    e.run()
                    x = describe(e.results(), q_mid)
                    if x: bad.append(x)
            finally:
                e.close()
        ts = [threading.Thread(target=worker) for _ in range(n)]
        [t.start() for t in ts]; [t.join() for t in ts]
        print(f"  (B) {n} engine(s) x 60 runs: {len(bad)} differ; examples {bad[:4]}", flush=True)
    # (C) time
    e = Engine("PNA", 0, options=opts(ab))
    e.set_weights(w); e.set_batch(big)
    for _ in range(3): e.run()
    e.sync(); e.profile_enable(True)
    for _ in range(8): e.run()
    e.sync()
    k = {a: round(v["total_ms"] / max(v["launches"], 1), 4) for a, v in e.profile_read().items() if "fused" in a}
    print(f"  (C) {k}", flush=True)
    e.close()
