"""Generates tests/golden/gat_molhiv48.npz (40 molhiv-shaped graphs + 8 with some reverse edges removed) with the
oracle's logits (per-graph feature offsets applied).  Parity unpinned: see make_gin_golden.py.
Run from the repo root:  python tests/golden/make_gat_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowgnn_amd import graphpack as gp, weights  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.test_oracle_gcn import directed_variant  # noqa: E402

b = gp.concat_batches([gp.synth_molhiv_batch(40, seed=20241001), directed_variant(gp.synth_molhiv_batch(8, seed=5))])
fields = dict(nums_of_nodes=b.nums_of_nodes, nums_of_edges=b.nums_of_edges, node_feature=b.node_feature,
              edge_list=b.edge_list, edge_attr=b.edge_attr,
              logits_synth_weights=oracle.gat_forward(b, [weights.synth_gat_weights(seed=7)]))
if os.path.isdir("/root/reference/GAT"):
    fields["logits_reference_weights"] = oracle.gat_forward(b, [weights.load_gat_weights("/root/reference/GAT")])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gat_molhiv48.npz"), **fields)
print("wrote gat_molhiv48.npz", fields["logits_synth_weights"][:4], fields.get("logits_reference_weights", [])[:4])
