"""Generates tests/golden/dgn_hep24.npz (16 hep10k-shaped kNN graphs + 8 molecule graphs, with eigenvector columns)
and the oracle's logits.  Parity unpinned: see make_gin_golden.py.
Run from the repo root:  python tests/golden/make_dgn_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowgnn_amd import graphpack as gp, weights  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.test_oracle_dgn import with_eigen  # noqa: E402

b = gp.concat_batches([gp.synth_hep10k_batch(16, seed=20240930), with_eigen(gp.synth_molhiv_batch(8, seed=5), 2)])
fields = dict(nums_of_nodes=b.nums_of_nodes, nums_of_edges=b.nums_of_edges, node_feature=b.node_feature,
              edge_list=b.edge_list, edge_attr=b.edge_attr, node_eigen=b.node_eigen,
              logits_synth_weights=oracle.dgn_forward(b, [weights.synth_dgn_weights(seed=7)]))
if os.path.isdir("/root/reference/DGN"):
    fields["logits_reference_weights"] = oracle.dgn_forward(b, [weights.load_dgn_weights("/root/reference/DGN")])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dgn_hep24.npz"), **fields)
print("wrote dgn_hep24.npz", fields["logits_synth_weights"][:4], fields.get("logits_reference_weights", [])[:4])
