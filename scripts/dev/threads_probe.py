"""Several host threads, each with engines of its own on ONE device, all six models at once: same logits as one after the other?"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from flowgnn_amd import Engine, compute_graphs, graphpack as gp, weights
MODELS = ["GIN", "GIN-VN", "GCN", "GAT", "PNA", "DGN"]
def job(model, seed):
    base = model.replace("-VN", "").lower()
    w = getattr(weights, f"synth_{base}_weights")(7)
    b = (gp.synth_hep10k_batch if model in ("PNA", "DGN") else gp.synth_molhiv_batch)(3000, seed=seed)
    if model == "GIN-VN":
        b = gp.add_virtual_nodes(b)
    return w, b
jobs = {m: job(m, 5 + i) for i, m in enumerate(MODELS)}
serial = {}
for m, (w, b) in jobs.items():
    e = Engine(m, 0); e.set_weights(w); serial[m] = e.forward(b).copy(); e.close()
res, errs = {}, []
def work(m, reps):
    try:
        w, b = jobs[m]
        e = Engine(m, 0); e.set_weights(w)
        for _ in range(reps):
            out = e.forward(b)
            ent = compute_graphs(m, b, [w])  # the entry points share a lock and their own engines
        res[m] = (out.copy(), ent.copy()); e.close()
    except Exception as ex:  # noqa
        errs.append((m, repr(ex)))
for rep in range(3):
    th = [threading.Thread(target=work, args=(m, 5)) for m in MODELS]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    for m in MODELS:
        out, ent = res[m]
        scale = max(1.0, float(np.abs(serial[m]).max()))
        exact = np.array_equal(out, serial[m])
        assert np.allclose(out, serial[m], rtol=1e-5, atol=1e-5 * scale) and np.allclose(ent, serial[m], rtol=1e-4, atol=1e-4 * scale), (m, np.abs(out - serial[m]).max())
        print(f"rep {rep} {m}: concurrent == serial bit for bit: {exact}", flush=True)
