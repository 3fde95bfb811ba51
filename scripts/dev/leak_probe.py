"""Create / use / destroy engines and groups repeatedly: does device memory come back?  usage: leak_probe.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from flowgnn_amd import Engine, EngineGroup, compute_graphs, graphpack as gp, weights
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
free0 = None
for r in range(rounds):
    for model in ("GIN", "GIN-VN", "GCN", "GAT", "PNA", "DGN"):
        base = model.replace("-VN", "").lower()
        w = getattr(weights, f"synth_{base}_weights")(7)
        b = (gp.synth_hep10k_batch if model in ("PNA", "DGN") else gp.synth_molhiv_batch)(200 + 37 * (r % 5), seed=r)
        if model == "GIN-VN":
            b = gp.add_virtual_nodes(b)
        e = Engine(model, 0); e.set_weights(w); o = e.forward(b); e.final_h(); e.close()
        g = EngineGroup(model, [0, 0]); g.set_weights(w); o2 = g.compute(b, 2); g.close()
        assert np.allclose(o, o2, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(o).max()))
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if r == 1:
        free0 = free
    if r in (1, rounds // 2, rounds - 1):
        print(f"round {r}: free {free / 2**20:.0f} MiB" + (f" (delta vs round 1: {(free - free0) / 2**20:+.1f} MiB)" if free0 else ""), flush=True)
