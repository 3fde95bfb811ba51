"""Generates tests/golden/modes.npz: the oracle's outputs for the two round-2 modes on the graphs of the existing fixtures
(gin_molhiv64, gcn_molpcba48, gat_molhiv48, pna_hep24, dgn_hep24 .npz) with the synthetic weights (seed 7):
  q_<model>   int16 bit patterns of the fixed-point mode (ap_fixed<16,6>; DGN ap_fixed<16,3>): ginq_oracle.c / q_oracle.c
  mt_<model>  [G][128] logits of the multi-task readout (GIN, GCN; NUM_TASK = 128 as ogbg-molpcba has)
Outputs of the ORACLE, not of the reference (parity unpinned: see make_gin_golden.py) -- they freeze the oracle and give the GPU
tests a fixed target.  Run from the repo root:  python tests/golden/make_modes_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from flowgnn_amd import graphpack as gp, weights  # noqa: E402
from oracle import oracle  # noqa: E402

FIXTURES = {"GIN": "gin_molhiv64", "GCN": "gcn_molpcba48", "GAT": "gat_molhiv48", "PNA": "pna_hep24", "DGN": "dgn_hep24"}
NUM_TASK = 128


def fixture_batch(model):
    z = np.load(os.path.join(HERE, FIXTURES[model] + ".npz"))
    eig = z["node_eigen"] if "node_eigen" in z.files else None
    return gp.GraphBatch(z["nums_of_nodes"], z["nums_of_edges"], z["node_feature"], z["edge_list"], z["edge_attr"], eig)


def expected():
    out = {}
    for model in FIXTURES:
        b = fixture_batch(model)
        w = weights.SYNTH[model](seed=7)
        if model == "GIN":
            _, pat = oracle.gin_forward_q(b, [w])
        else:
            _, pat = oracle.q_forward(model, b, [w])
        out["q_" + model.lower()] = np.asarray(pat, np.int16)
    for model, fwd in (("GIN", oracle.gin_forward), ("GCN", oracle.gcn_forward)):
        b = fixture_batch(model)
        w = weights.SYNTH[model](seed=7, num_tasks=NUM_TASK)
        out["mt_" + model.lower()] = np.asarray(fwd(b, [w], num_tasks=NUM_TASK), np.float32)
    return out


if __name__ == "__main__":
    f = expected()
    np.savez_compressed(os.path.join(HERE, "modes.npz"), **f)
    print("wrote modes.npz", {k: v.shape for k, v in f.items()})
