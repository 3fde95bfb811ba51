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
    names = re.findall(r"^\s*(?:const\s+char\*|int)\s+([A-Za-z_][A-Za-z0-9_]*)\s*\(", src, flags=re.M)
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
