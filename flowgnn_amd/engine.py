"""Host-side mirror of the reference's per-model entry points over the C ABI.

`Engine` wraps the handle API (flowgnn_create / set_weights / set_batch / run / get_results);
`compute_graphs` calls the reference-compatible `<M>_compute_graphs` symbol with host arrays,
exactly the argument list of the reference kernel (GIN/src/dcl.h:75-94).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np

from . import _lib
from .graphpack import GraphBatch


class FlowGNNError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        super().__init__(f"{where}: {_lib.STATUS.get(code, code)} {detail}".strip())


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _pi(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_lib.p_int)


def _pf(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_lib.p_float)


def option_value(v) -> float:
    """Option values are numbers; "f32" / "f16" (the *_mfma switches) read as 32 / 16."""
    if isinstance(v, str) and v in ("f32", "f16"):
        return 32.0 if v == "f32" else 16.0
    return float(v)


class Engine:
    """One engine = one GPU, one stream, one model's weights, one resident batch.
    `options`: {key: value} passed to flowgnn_set_option right after creation (e.g. {"gin_resident": 0})."""

    def __init__(self, model: str = "GIN", device: int = 0, options: Optional[Dict[str, float]] = None):
        self.lib = _lib.load()
        self.model = model.upper()
        if self.model not in _lib.MODEL_IDS:
            raise ValueError(f"unknown model {model}")
        self._h = C.c_void_p()
        self._check(self.lib.flowgnn_create(_lib.MODEL_IDS[self.model], device, C.byref(self._h)), "flowgnn_create")
        self._keep = []
        self.num_tasks = 1
        for k, v in (options or {}).items():
            self.set_option(k, v)

    def set_option(self, key: str, value):
        """Run-time switch by name (flowgnn.h: flowgnn_set_option); call before set_batch."""
        self._check(self.lib.flowgnn_set_option(self._h, key.encode(), option_value(value)), f"flowgnn_set_option({key})")

    def get_option(self, key: str) -> float:
        v = C.c_double()
        self._check(self.lib.flowgnn_get_option(self._h, key.encode(), C.byref(v)), f"flowgnn_get_option({key})")
        return float(v.value)

    def _check(self, rc: int, where: str):
        if rc != 0:
            detail = ""
            if self._h:
                detail = (self.lib.flowgnn_last_error(self._h) or b"").decode(errors="replace")
            raise FlowGNNError(rc, where, detail)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.flowgnn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights
    def set_weights(self, w: Dict[str, np.ndarray]):
        """`w`: OrderedDict in the argument order of the model's <M>_compute_graphs entry point."""
        arrs = [_f32(v) for v in w.values()]
        ptrs = (_lib.p_float * len(arrs))(*[_pf(a) for a in arrs])
        self._check(self.lib.flowgnn_set_weights(self._h, len(arrs), ptrs), "flowgnn_set_weights")

    def load_weights_dir(self, directory: str):
        self._check(self.lib.flowgnn_load_weights_dir(self._h, directory.encode()), "flowgnn_load_weights_dir")

    # ---- batch
    def set_job_totals(self, job_nodes: int = -1, job_edges: int = -1):
        """The next batches are shards of a job of this size (flowgnn.h: flowgnn_set_job_totals); (-1, -1): each batch is its own job."""
        self._check(self.lib.flowgnn_set_job_totals(self._h, int(job_nodes), int(job_edges)), "flowgnn_set_job_totals")

    def graph_tile_fill(self, nums_of_nodes, nums_of_edges) -> float:
        """Fill of this model's graph tiles for a graph list (flowgnn.h: flowgnn_graph_tile_fill); host code only."""
        nn, ne = _i32(nums_of_nodes), _i32(nums_of_edges)
        f = C.c_double()
        self._check(self.lib.flowgnn_graph_tile_fill(self._h, len(nn), _pi(nn), _pi(ne), C.byref(f)), "flowgnn_graph_tile_fill")
        return float(f.value)

    def batch_tiles(self):
        """(tiles in batch order, bin-packed tiles or 0) of the resident batch (flowgnn.h: flowgnn_batch_tiles)."""
        a, b = C.c_int(0), C.c_int(0)
        self._check(self.lib.flowgnn_batch_tiles(self._h, C.byref(a), C.byref(b)), "flowgnn_batch_tiles")
        return int(a.value), int(b.value)

    def set_job_tile_fill(self, fill: float = -1.0):
        """The next batches take the JOB's side of the resident kernels' fill threshold (flowgnn.h: flowgnn_set_job_tile_fill)."""
        self._check(self.lib.flowgnn_set_job_tile_fill(self._h, float(fill)), "flowgnn_set_job_tile_fill")

    def set_batch(self, batch: GraphBatch):
        nn, ne = _i32(batch.nums_of_nodes), _i32(batch.nums_of_edges)
        nf, el, ea = _i32(batch.node_feature), _i32(batch.edge_list), _i32(batch.edge_attr)
        eig = None if batch.node_eigen is None else _f32(batch.node_eigen)
        self._check(self.lib.flowgnn_set_batch(self._h, batch.num_graphs, _pi(nn), _pi(ne), _pi(nf), _pi(el), _pi(ea),
                                               _pf(eig)), "flowgnn_set_batch")
        self.num_graphs = batch.num_graphs
        self.total_nodes = batch.total_nodes
        self.total_edges = batch.total_edges

    def run(self):
        self._check(self.lib.flowgnn_run(self._h), "flowgnn_run")

    def sync(self):
        self._check(self.lib.flowgnn_sync(self._h), "flowgnn_sync")

    def results(self) -> np.ndarray:
        out = np.empty(self.num_graphs * self.num_tasks, dtype=np.float32)
        self._check(self.lib.flowgnn_get_results(self._h, _pf(out)), "flowgnn_get_results")
        return out.reshape(self.num_graphs, self.num_tasks) if self.num_tasks > 1 else out

    def set_num_tasks(self, num_tasks: int):
        """NUM_TASK of the readout (GIN / GIN-VN / GCN): set before weights and batch; results become [G][num_tasks]."""
        self._check(self.lib.flowgnn_set_num_tasks(self._h, int(num_tasks)), "flowgnn_set_num_tasks")
        self.num_tasks = int(num_tasks)

    def results_device_ptr(self) -> int:
        p = C.c_void_p()
        self._check(self.lib.flowgnn_results_device(self._h, C.byref(p)), "flowgnn_results_device")
        return int(p.value or 0)

    def set_results_buffer(self, device_ptr: int):
        self._check(self.lib.flowgnn_set_results_buffer(self._h, C.c_void_p(device_ptr)), "flowgnn_set_results_buffer")

    def forward(self, batch: GraphBatch) -> np.ndarray:
        self.set_batch(batch)
        self.run()
        return self.results()

    # ---- taps
    def set_numeric_mode(self, mode: str = "f32"):
        """"f32" (default) or "q6.10": the bit patterns of the reference's own fixed-point format (ap_fixed<16,6>; DGN: ap_fixed<16,3>)."""
        code = {"f32": 0, "q6.10": 1}[mode]
        self._check(self.lib.flowgnn_set_numeric_mode(self._h, code), "flowgnn_set_numeric_mode")

    def exact_reruns(self) -> int:
        """Forward passes repeated on the exact-fp32 kernels (flowgnn.h: flowgnn_exact_reruns)."""
        return int(self.lib.flowgnn_exact_reruns(self._h))

    def graph_replays(self) -> int:
        """Runs that were hipGraph replays of the recorded launch sequence (flowgnn.h: flowgnn_graph_replays)."""
        return int(self.lib.flowgnn_graph_replays(self._h))

    def csr(self):
        n, e = self.total_nodes, self.total_edges
        row_ptr = np.empty(n + 1, dtype=np.int32)
        src = np.empty(e, dtype=np.int32)
        eid = np.empty(e, dtype=np.int32)
        out_deg = np.empty(n, dtype=np.int32)
        self._check(self.lib.flowgnn_get_csr(self._h, _pi(row_ptr), _pi(src), _pi(eid), _pi(out_deg)), "flowgnn_get_csr")
        return row_ptr, src, eid, out_deg

    def final_h(self) -> np.ndarray:
        dim = C.c_int()
        self._check(self.lib.flowgnn_get_h(self._h, None, C.byref(dim)), "flowgnn_get_h")
        h = np.empty((self.total_nodes, dim.value), dtype=np.float32)
        self._check(self.lib.flowgnn_get_h(self._h, _pf(h), C.byref(dim)), "flowgnn_get_h")
        return h

    # ---- profiling
    def profile_enable(self, on: bool = True):
        self._check(self.lib.flowgnn_profile_enable(self._h, 1 if on else 0), "flowgnn_profile_enable")

    def profile_read(self) -> Dict[str, Dict[str, float]]:
        n = C.c_int()
        names = (C.c_char_p * 32)()
        ms = (C.c_double * 32)()
        cnt = (C.c_longlong * 32)()
        self._check(self.lib.flowgnn_profile_read(self._h, C.byref(n), names, ms, cnt), "flowgnn_profile_read")
        return {names[i].decode(): {"total_ms": ms[i], "launches": int(cnt[i])} for i in range(n.value)}

    def aggregation_only_ms(self, layer: int = 0, iters: int = 10) -> float:
        ms = C.c_float()
        self._check(self.lib.flowgnn_run_aggregation_only(self._h, layer, iters, C.byref(ms)),
                    "flowgnn_run_aggregation_only")
        return float(ms.value)


    def aggregate(self, layer: int = 0):
        """(rows read, aggregate written) by the standalone aggregation kernel of `layer` (flowgnn_get_aggregate)."""
        din, dagg = C.c_int(), C.c_int()
        self._check(self.lib.flowgnn_get_aggregate(self._h, layer, None, C.byref(din), None, C.byref(dagg)), "flowgnn_get_aggregate")
        h = np.empty((self.total_nodes, din.value), dtype=np.float32)
        a = np.empty((self.total_nodes, dagg.value), dtype=np.float32)
        self._check(self.lib.flowgnn_get_aggregate(self._h, layer, _pf(h), C.byref(din), _pf(a), C.byref(dagg)), "flowgnn_get_aggregate")
        return h, a

    def set_stream(self, stream_handle: Optional[int]):
        """Launch on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); None restores the engine's own."""
        if stream_handle is None:
            self._check(self.lib.flowgnn_set_stream(self._h, None, 0), "flowgnn_set_stream")
        else:
            self._check(self.lib.flowgnn_set_stream(self._h, C.c_void_p(stream_handle), 1), "flowgnn_set_stream")


class EngineGroup:
    """Several engines behind one handle (flowgnn.h: flowgnn_create_multi): the batch is cut into contiguous graph ranges
    balanced by sum(N + E), one engine + host thread per listed device; results come back in job order."""

    def __init__(self, model: str, devices, options: Optional[Dict[str, float]] = None):
        self.lib = _lib.load()
        self.model = model.upper()
        self.devices = [int(d) for d in devices]
        self._h = C.c_void_p()
        ids = _i32(self.devices)
        rc = self.lib.flowgnn_create_multi(_lib.MODEL_IDS[self.model], len(self.devices), _pi(ids), C.byref(self._h))
        if rc:
            raise FlowGNNError(rc, "flowgnn_create_multi")
        self.num_tasks = 1
        self.num_graphs = 0
        for k, v in (options or {}).items():
            self._check(self.lib.flowgnn_group_set_option(self._h, k.encode(), option_value(v)), f"flowgnn_group_set_option({k})")

    def _check(self, rc: int, where: str):
        if rc != 0:
            raise FlowGNNError(rc, where, (self.lib.flowgnn_group_last_error(self._h) or b"").decode(errors="replace"))

    def close(self):
        if getattr(self, "_h", None):
            self.lib.flowgnn_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_weights(self, w: Dict[str, np.ndarray]):
        arrs = [_f32(v) for v in w.values()]
        ptrs = (_lib.p_float * len(arrs))(*[_pf(a) for a in arrs])
        self._check(self.lib.flowgnn_group_set_weights(self._h, len(arrs), ptrs), "flowgnn_group_set_weights")

    def load_weights_dir(self, directory: str):
        self._check(self.lib.flowgnn_group_load_weights_dir(self._h, directory.encode()), "flowgnn_group_load_weights_dir")

    def set_num_tasks(self, num_tasks: int):
        self._check(self.lib.flowgnn_group_set_num_tasks(self._h, int(num_tasks)), "flowgnn_group_set_num_tasks")
        self.num_tasks = int(num_tasks)

    def set_numeric_mode(self, mode: str = "f32"):
        self._check(self.lib.flowgnn_group_set_numeric_mode(self._h, {"f32": 0, "q6.10": 1}[mode]), "flowgnn_group_set_numeric_mode")

    def set_batch(self, batch: GraphBatch):
        nn, ne = _i32(batch.nums_of_nodes), _i32(batch.nums_of_edges)
        nf, el, ea = _i32(batch.node_feature), _i32(batch.edge_list), _i32(batch.edge_attr)
        eig = None if batch.node_eigen is None else _f32(batch.node_eigen)
        self._check(self.lib.flowgnn_group_set_batch(self._h, batch.num_graphs, _pi(nn), _pi(ne), _pi(nf), _pi(el), _pi(ea), _pf(eig)),
                    "flowgnn_group_set_batch")
        self.num_graphs = batch.num_graphs

    def shards(self):
        cuts = np.zeros(len(self.devices) + 1, dtype=np.int32)
        self._check(self.lib.flowgnn_group_shards(self._h, _pi(cuts)), "flowgnn_group_shards")
        return [(int(cuts[i]), int(cuts[i + 1])) for i in range(len(self.devices))]

    def run(self):
        self._check(self.lib.flowgnn_group_run(self._h), "flowgnn_group_run")

    def sync(self):
        self._check(self.lib.flowgnn_group_sync(self._h), "flowgnn_group_sync")

    def results(self) -> np.ndarray:
        out = np.empty(self.num_graphs * self.num_tasks, dtype=np.float32)
        self._check(self.lib.flowgnn_group_get_results(self._h, _pf(out)), "flowgnn_group_get_results")
        return out.reshape(self.num_graphs, self.num_tasks) if self.num_tasks > 1 else out

    def forward(self, batch: GraphBatch) -> np.ndarray:
        self.set_batch(batch)
        self.run()
        return self.results()

    def compute(self, batch: GraphBatch, chunks_per_engine: int = 1) -> np.ndarray:
        """flowgnn_group_compute: the host batch cut into size x chunks_per_engine ranges, engine i taking ranges i, i + size, ...
        (set_batch, run, results) so that one engine's host -> device copies overlap the others' kernels.  The engines are left on
        their last range: run() / results() / shards() raise FLOWGNN_ERR_STATE until the next set_batch."""
        nn, ne = _i32(batch.nums_of_nodes), _i32(batch.nums_of_edges)
        nf, el, ea = _i32(batch.node_feature), _i32(batch.edge_list), _i32(batch.edge_attr)
        eig = None if batch.node_eigen is None else _f32(batch.node_eigen)
        out = np.empty(batch.num_graphs * self.num_tasks, dtype=np.float32)
        self._check(self.lib.flowgnn_group_compute(self._h, batch.num_graphs, _pi(nn), _pi(ne), _pi(nf), _pi(el), _pi(ea), _pf(eig), _pf(out),
                                                   int(chunks_per_engine)), "flowgnn_group_compute")
        return out.reshape(batch.num_graphs, self.num_tasks) if self.num_tasks > 1 else out


def shard_ranges_c(nums_of_nodes, nums_of_edges, parts: int):
    """flowgnn_shard_ranges (the C ABI's cut by cumulative node + edge count); pure host code, no GPU needed."""
    lib = _lib.load()
    nn, ne = _i32(nums_of_nodes), _i32(nums_of_edges)
    cuts = np.zeros(parts + 1, dtype=np.int32)
    rc = lib.flowgnn_shard_ranges(len(nn), _pi(nn), _pi(ne), parts, _pi(cuts))
    if rc:
        raise FlowGNNError(rc, "flowgnn_shard_ranges")
    return [(int(cuts[i]), int(cuts[i + 1])) for i in range(parts)]


def entry_set_devices(devices):
    """Devices of the <M>_compute_graphs entry points (flowgnn.h: flowgnn_entry_set_devices)."""
    ids = _i32(list(devices))
    rc = _lib.load().flowgnn_entry_set_devices(len(ids), _pi(ids))
    if rc:
        raise FlowGNNError(rc, "flowgnn_entry_set_devices")


def entry_set_pipeline(chunks_per_engine: int):
    """Ranges per engine of the entry points' host-array pipeline (flowgnn.h: flowgnn_entry_set_pipeline; 0 = by size, 1 = off)."""
    rc = _lib.load().flowgnn_entry_set_pipeline(int(chunks_per_engine))
    if rc:
        raise FlowGNNError(rc, "flowgnn_entry_set_pipeline")


def entry_set_option(model: str, key: str, value):
    rc = _lib.load().flowgnn_entry_set_option(_lib.MODEL_IDS[model.upper()], key.encode(), option_value(value))
    if rc:
        raise FlowGNNError(rc, f"flowgnn_entry_set_option({key})")


def compute_graphs(model: str, batch: GraphBatch, weight_sets, reload_weights=None, num_tasks: int = 1) -> np.ndarray:
    """Call the reference-compatible C symbol <M>_compute_graphs (e.g. GIN/src/dcl.h:75-94) with host
    arrays.  `weight_sets` is a list of weight dicts: the leading [S] dimension of every weight pointer,
    selected per graph by the running count of reload_weights (GIN/src/GIN_compute.cc:51-53)."""
    lib = _lib.load()
    model = model.upper()
    G = batch.num_graphs
    if reload_weights is None:
        reload_weights = np.zeros(G, dtype=np.int32)
        if G:
            reload_weights[0] = 1
    stacked = [np.ascontiguousarray(np.stack([np.asarray(ws[k], dtype=np.float32) for ws in weight_sets]))
               for k in weight_sets[0].keys()]  # [S, ...] per tensor (avg_deg: [S, 1] == float[S])
    out = np.zeros(G * num_tasks, dtype=np.float32)
    if num_tasks != 1 and model not in ("GIN", "GIN-VN", "GCN"):
        raise FlowGNNError(8, f"{model}_compute_graphs", "this model's readout is a single-task MLP head (NUM_TASK != 1 exists for GIN / GIN-VN / GCN)")
    nn, ne, rw = _i32(batch.nums_of_nodes), _i32(batch.nums_of_edges), _i32(reload_weights)
    nf, el, ea = _i32(batch.node_feature), _i32(batch.edge_list), _i32(batch.edge_attr)
    wp = [_pf(a) for a in stacked]
    if model in ("GIN", "GIN-VN"):
        if num_tasks == 1:
            rc = lib.GIN_compute_graphs(G, _pi(nn), _pi(ne), _pi(rw), _pf(out), _pi(nf), _pi(el), _pi(ea), *wp)
        else:  # NUM_TASK, a compile-time constant of the reference build, is an explicit argument here
            rc = lib.GIN_compute_graphs_mt(G, _pi(nn), _pi(ne), _pi(rw), _pf(out), _pi(nf), _pi(el), _pi(ea), *wp, num_tasks)
    elif model == "GCN":
        if num_tasks == 1:
            rc = lib.GCN_compute_graphs(G, _pi(nn), _pi(ne), _pi(rw), _pf(out), _pi(nf), _pi(el), _pi(ea), *wp)
        else:
            rc = lib.GCN_compute_graphs_mt(G, _pi(nn), _pi(ne), _pi(rw), _pf(out), _pi(nf), _pi(el), _pi(ea), *wp, num_tasks)
    elif model == "PNA":
        rc = lib.PNA_compute_graphs(G, _pi(nn), _pi(ne), _pi(rw), _pf(out), _pi(nf), _pi(el), *wp)
    elif model == "GAT":
        rc = lib.GAT_compute_graphs(G, _pi(nn), _pi(ne), _pi(rw), _pf(out), _pi(nf), _pi(el), *wp)
    elif model == "DGN":
        eig = _f32(batch.node_eigen)
        rc = lib.DGN_compute_graphs(G, _pi(nn), _pi(ne), _pi(rw), _pf(out), _pi(nf), _pf(eig), _pi(el), *wp)
    else:
        raise ValueError(model)
    if rc:
        raise FlowGNNError(rc, f"{model}_compute_graphs")
    return out.reshape(G, num_tasks) if num_tasks > 1 else out


def GIN_compute_graphs(batch: GraphBatch, weight_sets, reload_weights=None) -> np.ndarray:
    return compute_graphs("GIN", batch, weight_sets, reload_weights)


def GCN_compute_graphs(batch: GraphBatch, weight_sets, reload_weights=None) -> np.ndarray:
    return compute_graphs("GCN", batch, weight_sets, reload_weights)
