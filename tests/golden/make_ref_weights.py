"""Generates tests/golden/ref_weights_<model>.npz: the TRAINED weight sets the reference ships as raw float32 .bin files
(GIN/*.bin, GCN/gcn_ep1_dim100.weights.all.bin, GAT/*.bin, PNA/pna_ep1_noBN_dim80.weights.all.bin, DGN/dgn_ep1_noBN_dim100.weights.all.bin;
GIN-VN ships the same files as GIN), read with this repo's loaders (flowgnn_amd/weights.py, which follow <M>/src/host_load.cc) and
stored tensor by tensor.  Data, not code: the GPU box has no /root/reference, and parity on trained weights (magnitudes the
synthetic sets do not have: GIN logits around -3 .. -6, PNA's saturating head) is otherwise only ever checked on the CPU.
The expected logits for them are the `logits_reference_weights` arrays of the per-model fixtures next to this file (oracle outputs).
Run from the repo root, where /root/reference exists:  python tests/golden/make_ref_weights.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from flowgnn_amd import weights  # noqa: E402

REF = os.environ.get("FLOWGNN_REFERENCE", "/root/reference")

if __name__ == "__main__":
    for model in ("GIN", "GCN", "GAT", "PNA", "DGN"):
        w = weights.LOADERS[model](os.path.join(REF, model))
        path = os.path.join(HERE, f"ref_weights_{model.lower()}.npz")
        np.savez_compressed(path, **{k: np.asarray(v, np.float32) for k, v in w.items()})
        print(model, {k: v.shape for k, v in w.items()}, os.path.getsize(path), "bytes")
