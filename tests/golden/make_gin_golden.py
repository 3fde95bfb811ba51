"""Generates tests/golden/gin_molhiv64.npz: 64 molhiv-shaped synthetic graphs (inputs) and the
oracle's outputs for them -- with seeded synthetic weights (reproducible anywhere) and, when
/root/reference is present, with the reference's shipped GIN weights.

NOTE (parity unpinned): these vectors are outputs of oracle/gin_oracle.c, not of the reference
itself; the reference kernel cannot be built in this image (no Vitis HLS headers) and ships no
golden outputs.  They pin the oracle against regressions and give the GPU tests a fixed target.

Run from the repo root:  python tests/golden/make_gin_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowgnn_amd import graphpack as gp, weights  # noqa: E402
from oracle import oracle  # noqa: E402

b = gp.synth_molhiv_batch(64, seed=20240928)
w = weights.synth_gin_weights(seed=7)
out, hd = oracle.gin_forward(b, [w], dump_h=True)
n4 = int(b.nums_of_nodes[:4].sum())
fields = dict(nums_of_nodes=b.nums_of_nodes, nums_of_edges=b.nums_of_edges, node_feature=b.node_feature,
              edge_list=b.edge_list, edge_attr=b.edge_attr, logits_synth_weights=out,
              h_first4_graphs=hd[:, :n4].copy())
if os.path.isdir("/root/reference/GIN"):
    fields["logits_reference_weights"] = oracle.gin_forward(b, [weights.load_gin_weights("/root/reference/GIN")])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gin_molhiv64.npz"), **fields)
print("wrote gin_molhiv64.npz", out[:4])
