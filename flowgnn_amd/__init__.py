"""flowgnn_amd -- MI355X-native engine for FlowGNN's NT/MP inference hot path.

The compute lives in libflowgnn_hip.so (hand-written HIP for gfx950 behind the C ABI of
include/flowgnn.h).  This package is the thin host-side mirror used by tests and bench.py;
importing it does not load the library, using an Engine does, and fails loudly if it is absent.
"""
from .graphpack import (GraphBatch, synth_molhiv_batch, synth_molpcba_batch, synth_hep10k_batch,  # noqa: F401
                        add_virtual_nodes, read_pack, write_pack, concat_batches)
from .engine import (Engine, EngineGroup, FlowGNNError, compute_graphs, GIN_compute_graphs, GCN_compute_graphs,  # noqa: F401
                     entry_set_devices, entry_set_option, entry_set_pipeline, shard_ranges_c)
from . import weights  # noqa: F401

__all__ = ["Engine", "EngineGroup", "FlowGNNError", "GIN_compute_graphs", "GraphBatch", "weights"]
