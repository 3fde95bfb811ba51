"""Weight sets in the reference's raw little-endian float32 `.bin` layouts (no header),
plus seeded synthetic weights with the same shapes and scales.

Readers follow the reference host loaders:
  GIN   nine separate files                         GIN/src/host_load.cc:24-58
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict

import numpy as np

# name -> (file, shape); order = argument order of flowgnn_set_weights_gin / GIN_compute_graphs
GIN_FILES = OrderedDict([
    ("node_embedding_weight", ("gin_ep1_nd_embed_dim100.bin", (173, 100))),
    ("edge_embedding_weight", ("gin_ep1_ed_embed_dim100.bin", (5, 13, 100))),
    ("node_mlp_1_weights", ("gin_ep1_mlp_1_weights_dim100.bin", (5, 200, 100))),
    ("node_mlp_1_bias", ("gin_ep1_mlp_1_bias_dim100.bin", (5, 200))),
    ("node_mlp_2_weights", ("gin_ep1_mlp_2_weights_dim100.bin", (5, 100, 200))),
    ("node_mlp_2_bias", ("gin_ep1_mlp_2_bias_dim100.bin", (5, 100))),
    ("graph_pred_weights", ("gin_ep1_pred_weights_dim100.bin", (1, 100))),
    ("graph_pred_bias", ("gin_ep1_pred_bias_dim100.bin", (1,))),
])


def _read(path: str, shape, offset_floats: int = 0) -> np.ndarray:
    count = int(np.prod(shape))
    arr = np.fromfile(path, dtype="<f4", count=count, offset=4 * offset_floats)
    if arr.size != count:
        raise IOError(f"{path}: wanted {count} floats at offset {offset_floats}, got {arr.size}")
    return np.ascontiguousarray(arr.reshape(shape), dtype=np.float32)


def load_gin_weights(directory: str) -> Dict[str, np.ndarray]:
    return OrderedDict((k, _read(os.path.join(directory, f), shp)) for k, (f, shp) in GIN_FILES.items())


def save_gin_weights(w: Dict[str, np.ndarray], directory: str) -> None:
    os.makedirs(directory, exist_ok=True)
    for k, (f, shp) in GIN_FILES.items():
        np.asarray(w[k], dtype="<f4").reshape(shp).tofile(os.path.join(directory, f))
    np.zeros(5, dtype="<f4").tofile(os.path.join(directory, "gin_ep1_eps_dim100.bin"))  # read, never used


def synth_gin_weights(seed: int = 7) -> Dict[str, np.ndarray]:
    """Random weights with the measured scales of the shipped GIN set:
    W1 s=0.075, b1 s=0.91, W2 s=0.053, b2 s=0.49, node/edge emb s=0.11/0.14, pred s=0.15."""
    rng = np.random.default_rng(seed)
    scale = {
        "node_embedding_weight": 0.11, "edge_embedding_weight": 0.14,
        "node_mlp_1_weights": 0.075, "node_mlp_1_bias": 0.91,
        "node_mlp_2_weights": 0.053, "node_mlp_2_bias": 0.49,
        "graph_pred_weights": 0.15, "graph_pred_bias": 0.12,
    }
    return OrderedDict((k, (rng.standard_normal(shp) * scale[k]).astype(np.float32))
                       for k, (_, shp) in GIN_FILES.items())
