"""ctypes binding of libflowgnn_hip.so (the C ABI of include/flowgnn.h).

There is no CPU fallback: if the HIP library is missing this module raises on load.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libflowgnn_hip.so")

STATUS = {
    0: "FLOWGNN_OK", 1: "FLOWGNN_ERR_ARG", 2: "FLOWGNN_ERR_EDGE_RANGE", 3: "FLOWGNN_ERR_EDGE_ATTR",
    4: "FLOWGNN_ERR_NODE_FEAT", 5: "FLOWGNN_ERR_HIP", 6: "FLOWGNN_ERR_STATE", 7: "FLOWGNN_ERR_IO",
    8: "FLOWGNN_ERR_UNSUPPORTED",
}

MODEL_IDS = {"GIN": 0, "GIN-VN": 1, "GCN": 2, "GAT": 3, "PNA": 4, "DGN": 5}

_lib = None

p_int = C.POINTER(C.c_int)
p_float = C.POINTER(C.c_float)
p_void = C.c_void_p


def _declare(lib):
    eng = C.c_void_p
    lib.flowgnn_create.argtypes = [C.c_int, C.c_int, C.POINTER(eng)]
    lib.flowgnn_destroy.argtypes = [eng]
    lib.flowgnn_last_error.argtypes = [eng]
    lib.flowgnn_last_error.restype = C.c_char_p
    lib.flowgnn_set_weights_gin.argtypes = [eng] + [p_float] * 8
    lib.flowgnn_set_weights.argtypes = [eng, C.c_int, C.POINTER(p_float)]
    lib.flowgnn_load_weights_dir.argtypes = [eng, C.c_char_p]
    lib.flowgnn_set_batch.argtypes = [eng, C.c_int, p_int, p_int, p_int, p_int, p_int, p_float]
    lib.flowgnn_set_job_totals.argtypes = [eng, C.c_longlong, C.c_longlong]
    lib.flowgnn_run.argtypes = [eng]
    lib.flowgnn_sync.argtypes = [eng]
    lib.flowgnn_get_results.argtypes = [eng, p_float]
    lib.flowgnn_results_device.argtypes = [eng, C.POINTER(C.c_void_p)]
    lib.flowgnn_set_results_buffer.argtypes = [eng, C.c_void_p]
    lib.flowgnn_stream.argtypes = [eng, C.POINTER(C.c_void_p)]
    lib.flowgnn_batch_info.argtypes = [eng] + [C.POINTER(C.c_longlong)] * 3
    lib.flowgnn_batch_tiles.argtypes = [eng] + [C.POINTER(C.c_int)] * 2
    lib.flowgnn_exact_reruns.argtypes = [eng]
    lib.flowgnn_graph_replays.argtypes = [eng]
    lib.flowgnn_graph_replays.restype = C.c_longlong
    lib.flowgnn_set_numeric_mode.argtypes = [eng, C.c_int]
    lib.flowgnn_set_num_tasks.argtypes = [eng, C.c_int]
    lib.flowgnn_num_tasks.argtypes = [eng]
    lib.flowgnn_get_csr.argtypes = [eng, p_int, p_int, p_int, p_int]
    lib.flowgnn_get_h.argtypes = [eng, p_float, p_int]
    lib.flowgnn_profile_enable.argtypes = [eng, C.c_int]
    lib.flowgnn_profile_read.argtypes = [eng, p_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                         C.POINTER(C.c_longlong)]
    lib.flowgnn_run_aggregation_only.argtypes = [eng, C.c_int, C.c_int, p_float]
    lib.flowgnn_get_aggregate.argtypes = [eng, C.c_int, p_float, p_int, p_float, p_int]
    lib.flowgnn_set_stream.argtypes = [eng, C.c_void_p, C.c_int]
    grp = C.c_void_p
    lib.flowgnn_set_option.argtypes = [eng, C.c_char_p, C.c_double]
    lib.flowgnn_get_option.argtypes = [eng, C.c_char_p, C.POINTER(C.c_double)]
    lib.flowgnn_option_count.argtypes = []
    lib.flowgnn_option_name.argtypes = [C.c_int]
    lib.flowgnn_option_name.restype = C.c_char_p
    lib.flowgnn_entry_set_devices.argtypes = [C.c_int, p_int]
    lib.flowgnn_entry_set_option.argtypes = [C.c_int, C.c_char_p, C.c_double]
    lib.flowgnn_entry_set_pipeline.argtypes = [C.c_int]
    lib.flowgnn_shard_ranges.argtypes = [C.c_int, p_int, p_int, C.c_int, p_int]
    lib.flowgnn_graph_tile_fill.argtypes = [eng, C.c_int, p_int, p_int, C.POINTER(C.c_double)]
    lib.flowgnn_set_job_tile_fill.argtypes = [eng, C.c_double]
    lib.flowgnn_create_multi.argtypes = [C.c_int, C.c_int, p_int, C.POINTER(grp)]
    lib.flowgnn_group_destroy.argtypes = [grp]
    lib.flowgnn_group_size.argtypes = [grp]
    lib.flowgnn_group_engine.argtypes = [grp, C.c_int]
    lib.flowgnn_group_engine.restype = C.c_void_p
    lib.flowgnn_group_last_error.argtypes = [grp]
    lib.flowgnn_group_last_error.restype = C.c_char_p
    lib.flowgnn_group_set_weights.argtypes = [grp, C.c_int, C.POINTER(p_float)]
    lib.flowgnn_group_load_weights_dir.argtypes = [grp, C.c_char_p]
    lib.flowgnn_group_set_option.argtypes = [grp, C.c_char_p, C.c_double]
    lib.flowgnn_group_set_num_tasks.argtypes = [grp, C.c_int]
    lib.flowgnn_group_set_numeric_mode.argtypes = [grp, C.c_int]
    lib.flowgnn_group_set_batch.argtypes = [grp, C.c_int, p_int, p_int, p_int, p_int, p_int, p_float]
    lib.flowgnn_group_shards.argtypes = [grp, p_int]
    lib.flowgnn_group_run.argtypes = [grp]
    lib.flowgnn_group_sync.argtypes = [grp]
    lib.flowgnn_group_get_results.argtypes = [grp, p_float]
    lib.flowgnn_group_compute.argtypes = [grp, C.c_int, p_int, p_int, p_int, p_int, p_int, p_float, p_float, C.c_int]
    lib.GIN_compute_graphs_mt.argtypes = [C.c_int, p_int, p_int, p_int, p_float, p_int, p_int, p_int] + [p_float] * 8 + [C.c_int]
    lib.GCN_compute_graphs_mt.argtypes = [C.c_int, p_int, p_int, p_int, p_float, p_int, p_int, p_int] + [p_float] * 11 + [C.c_int]
    lib.GIN_compute_graphs.argtypes = [C.c_int, p_int, p_int, p_int, p_float, p_int, p_int, p_int] + [p_float] * 8
    lib.PNA_compute_graphs.argtypes = [C.c_int, p_int, p_int, p_int, p_float, p_int, p_int] + [p_float] * 10
    lib.DGN_compute_graphs.argtypes = [C.c_int, p_int, p_int, p_int, p_float, p_int, p_float, p_int] + [p_float] * 9
    lib.GAT_compute_graphs.argtypes = [C.c_int, p_int, p_int, p_int, p_float, p_int, p_int] + [p_float] * 6
    lib.GCN_compute_graphs.argtypes = [C.c_int, p_int, p_int, p_int, p_float, p_int, p_int, p_int] + [p_float] * 11
    for name in ("flowgnn_create", "flowgnn_destroy", "flowgnn_set_weights_gin", "flowgnn_set_weights", "flowgnn_load_weights_dir",
                 "flowgnn_set_batch", "flowgnn_set_job_totals", "flowgnn_graph_tile_fill", "flowgnn_set_job_tile_fill", "flowgnn_run", "flowgnn_sync", "flowgnn_get_results",
                 "flowgnn_results_device", "flowgnn_set_results_buffer", "flowgnn_stream", "flowgnn_batch_info", "flowgnn_batch_tiles", "flowgnn_exact_reruns", "flowgnn_set_numeric_mode", "flowgnn_set_num_tasks", "flowgnn_num_tasks", "flowgnn_get_csr",
                 "flowgnn_get_h", "flowgnn_profile_enable", "flowgnn_profile_read",
                 "flowgnn_run_aggregation_only", "flowgnn_get_aggregate", "flowgnn_set_stream",
                 "flowgnn_set_option", "flowgnn_get_option", "flowgnn_option_count", "flowgnn_entry_set_devices", "flowgnn_entry_set_option",
                 "flowgnn_shard_ranges", "flowgnn_create_multi", "flowgnn_group_destroy", "flowgnn_group_size", "flowgnn_group_set_weights",
                 "flowgnn_group_load_weights_dir", "flowgnn_group_set_option", "flowgnn_group_set_num_tasks", "flowgnn_group_set_numeric_mode",
                 "flowgnn_group_set_batch", "flowgnn_group_shards", "flowgnn_group_run", "flowgnn_group_sync", "flowgnn_group_get_results",
                 "flowgnn_group_compute", "flowgnn_entry_set_pipeline",
                 "GIN_compute_graphs_mt", "GCN_compute_graphs_mt", "GIN_compute_graphs", "GCN_compute_graphs", "PNA_compute_graphs", "DGN_compute_graphs", "GAT_compute_graphs"):
        getattr(lib, name).restype = C.c_int


def load():
    """Load libflowgnn_hip.so (built by `python -c 'import __graft_entry__ as g; g.build()'`
    or `make -C flowgnn_amd/csrc`).  Raises if it is missing: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built. "
                "Run `make -C flowgnn_amd/csrc` (needs hipcc, --offload-arch=gfx950).")
        lib = C.CDLL(LIB_PATH)
        _declare(lib)
        _lib = lib
    return _lib
