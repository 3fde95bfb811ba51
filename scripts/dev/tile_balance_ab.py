#!/usr/bin/env python3
"""A/B of option tile_balance (flowgnn_set_batch: whole rounds of smaller graph tiles) on the dataset-sized batches of every BASELINE
config: ms per step, 500 timed steps behind 300 warm-up steps, no profiling.  usage: python scripts/dev/tile_balance_ab.py [models...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from flowgnn_amd import Engine, graphpack as gp, weights

CASES = {"GIN": ("molhiv", 4113), "GIN-VN": ("molhiv-vn", 4113), "GCN": ("molpcba", 43773), "GAT": ("molhiv", 4113), "PNA": ("hep10k", 10000),
         "DGN": ("hep10k", 10000)}


def batch(kind, n):
    if kind == "molhiv":
        return gp.synth_molhiv_batch(n, seed=99)
    if kind == "molhiv-vn":
        return gp.add_virtual_nodes(gp.synth_molhiv_batch(n, seed=99))
    if kind == "molpcba":
        return gp.synth_molpcba_batch(n, seed=99)
    return gp.synth_hep10k_batch(n, seed=99)


for m in (sys.argv[1:] or list(CASES)):
    b = batch(*CASES[m])
    w = weights.SYNTH[m](seed=7)
    res, outs = {}, {}
    for rep in range(2):
        for tb in (0, 1):
            e = Engine(m, device=0, options={"tile_balance": tb})
            e.set_weights(w)
            e.set_batch(b)
            for _ in range(300):
                e.run()
            e.sync()
            t0 = time.perf_counter()
            for _ in range(500):
                e.run()
            e.sync()
            res.setdefault(tb, []).append((time.perf_counter() - t0) / 500 * 1e3)
            outs[tb] = e.results()
            e.close()
    same = np.array_equal(outs[0], outs[1])
    print(f"{m:7s} {CASES[m][1]:6d} graphs  off {min(res[0]):.4f} ms  on {min(res[1]):.4f} ms  ({(min(res[1]) / min(res[0]) - 1) * 100:+.1f} %)  "
          f"{b.num_graphs / min(res[1]) / 1e3:.2f} M graphs/s  bits {'same' if same else 'differ %.2e' % np.abs(outs[0] - outs[1]).max()}", flush=True)
