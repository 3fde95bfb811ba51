"""The C-ABI library loads and exports every symbol include/flowgnn.h declares (no compute calls:
this runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "flowgnn_amd", "libflowgnn_hip.so")
HEADER = os.path.join(ROOT, "include", "flowgnn.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+char\*|flowgnn_engine\*|int)\s+([A-Za-z_][A-Za-z0-9_]*)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_header_declares_entry_points():
    names = declared_functions()
    assert "GIN_compute_graphs" in names
    assert "flowgnn_create" in names and "flowgnn_run" in names


@pytest.mark.skipif(not os.path.exists(LIB), reason="libflowgnn_hip.so not built (run __graft_entry__.build())")
def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(LIB)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


@pytest.mark.skipif(not os.path.exists(LIB), reason="libflowgnn_hip.so not built")
def test_bad_arguments_are_rejected_without_a_gpu():
    lib = ctypes.CDLL(LIB)
    assert lib.flowgnn_create(0, 0, None) == 1          # FLOWGNN_ERR_ARG
    assert lib.flowgnn_destroy(None) == 1
    assert lib.GIN_compute_graphs(-1, *([None] * 15)) == 1
    assert lib.GIN_compute_graphs(0, *([None] * 15)) == 0  # empty batch is a no-op
    lib.flowgnn_set_job_totals.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong]
    assert lib.flowgnn_set_job_totals(None, 10, 20) == 1
    assert lib.flowgnn_group_run(None) == 1 and lib.flowgnn_group_get_results(None, None) == 1


@pytest.mark.skipif(not os.path.exists(LIB), reason="libflowgnn_hip.so not built")
def test_shipped_library_has_no_ablation_hooks_and_one_env_reader():
    """The kernels' ablation hooks (some give wrong results on purpose) exist only in a -DFLOWGNN_DEV build; NUM_TASK is an
    argument, not an environment variable; every option name is reachable through the API."""
    blob = open(LIB, "rb").read()
    assert b"ABLATE" not in blob and b"_ablate" not in blob
    # FLOWGNN_NUM_TASK appears once: as the name the plain GIN / GCN entry points REFUSE to run under (a stale setting of round 2's
    # switch would otherwise give one task's results for [T][100] weights silently)
    assert blob.count(b"FLOWGNN_NUM_TASK is set in the environment but no longer read") == 1
    lib = ctypes.CDLL(LIB)
    lib.flowgnn_option_name.restype = ctypes.c_char_p
    names = [lib.flowgnn_option_name(i).decode() for i in range(lib.flowgnn_option_count())]
    assert "gin_resident" in names and "hipgraph" in names and "pna_fused" in names
    assert not [n for n in names if n.endswith("ablate")]
    assert lib.flowgnn_option_name(len(names)) is None
    # the library reads its environment in one function: every getenv call site sits in engine.hip's read_environment
    import glob
    hits = []
    for f in glob.glob(os.path.join(ROOT, "flowgnn_amd", "csrc", "*")):
        if f.endswith((".hip", ".h", ".cpp")):
            for i, line in enumerate(open(f, errors="replace"), 1):
                code = line.split("//")[0]
                if "getenv(" in code:
                    hits.append((os.path.basename(f), i))
    assert hits and all(f == "engine.hip" for f, _ in hits) and len(hits) <= 4, hits


@pytest.mark.skipif(not os.path.exists(LIB), reason="libflowgnn_hip.so not built")
def test_shard_ranges_c_equals_python():
    """flowgnn_shard_ranges (host code of the multi-device path) cuts exactly where flowgnn_amd.dist.shard_ranges does."""
    import numpy as np
    from flowgnn_amd import graphpack as gp, shard_ranges_c
    from flowgnn_amd.dist import shard_ranges
    rng = np.random.default_rng(0)
    for trial in range(20):
        G = int(rng.integers(1, 200))
        b = gp.synth_hep10k_batch(G, seed=trial, with_eigen=False) if trial % 2 else gp.synth_molhiv_batch(G, seed=trial)
        for parts in (1, 2, 3, 8, 17, 256):
            assert shard_ranges_c(b.nums_of_nodes, b.nums_of_edges, parts) == shard_ranges(b, parts), (trial, parts)
    assert shard_ranges_c(np.zeros(0, np.int32), np.zeros(0, np.int32), 4) == [(0, 0)] * 4


@pytest.mark.skipif(not os.path.exists(LIB), reason="libflowgnn_hip.so not built")
def test_stale_num_task_refusal_fills_the_output_and_clears_when_unset():
    """The plain GIN / GCN symbols are `void` in the reference (GIN/src/dcl.h:75-94): a caller that ignores the status must not read an
    untouched buffer as results.  A refusal fills `out` with NaN, and unsetting the variable in the same process clears it."""
    import numpy as np
    lib = ctypes.CDLL(LIB)
    out = np.ones(4, np.float32)
    args = [None] * 15
    args[3] = out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    os.environ["FLOWGNN_NUM_TASK"] = "3"
    try:
        for sym in ("GIN_compute_graphs", "GCN_compute_graphs"):
            out[:] = 1.0
            assert getattr(lib, sym)(4, *args) == 8  # FLOWGNN_ERR_UNSUPPORTED
            assert np.isnan(out).all()
    finally:
        del os.environ["FLOWGNN_NUM_TASK"]
    assert lib.GIN_compute_graphs(0, *([None] * 15)) == 0  # the same process, variable unset: no refusal (empty batch: no device needed)
    os.environ["FLOWGNN_NUM_TASK"] = "1"  # the default value is not stale
    try:
        assert lib.GCN_compute_graphs(0, *([None] * 15)) == 0
    finally:
        del os.environ["FLOWGNN_NUM_TASK"]
