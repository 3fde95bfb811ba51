"""oracle/q_oracle.c (GCN / GAT / PNA in ap_fixed<16,6>, DGN in ap_fixed<16,3>): internal consistency of the rules it assumes
(R0..R8 in its header), and the magnitudes it produces with the shipped trained weights when /root/reference is present
(SURVEY 8c observed the reference's own kernels, run with a throw-away header stand-in, at GCN ~ -2.4..-3.2, GAT ~ -0.2..-1.4,
PNA ~ -1.2..-1.4, DGN ~ -0.40..-0.43 on its synthetic graphs)."""
import ctypes as C
import os

import numpy as np
import pytest

from flowgnn_amd import graphpack as gp, weights
from tests.test_oracle_dgn import with_eigen

REF = "/root/reference"


def test_function_tables(oracle):
    lib = oracle.load()
    lib.orc_q_exp_table.restype = C.POINTER(C.c_int16)
    lib.orc_q_from_float.restype = C.c_int16
    lib.orc_q_from_float.argtypes = [C.c_float, C.c_int]
    lib.orc_q_log.restype = C.c_int16
    lib.orc_q_log.argtypes = [C.c_int16]
    t = lib.orc_q_exp_table()
    assert t[0] == 1024                       # exp(0) = 1
    assert t[1024] == int(np.floor(np.e * 1024))
    assert t[(-1024) & 0xFFFF] == int(np.floor(np.exp(-1.0) * 1024))
    assert t[(-32768) & 0xFFFF] == 0          # exp(-32) floors to 0 on the grid
    grid = [t[p] for p in range(0, 3 * 1024)]  # below ln(32) the table is monotone and has not wrapped
    assert all(b >= a for a, b in zip(grid, grid[1:])) and max(grid) < 32768
    assert lib.orc_q_from_float(0.2, 10) == 204 and lib.orc_q_from_float(-0.2, 10) == -205   # floor, not round
    assert lib.orc_q_from_float(1.0, 13) == 8192 and lib.orc_q_from_float(4.0, 13) == -32768  # wrap at +4 in Q3.13
    assert lib.orc_q_log(1024) == 0 and lib.orc_q_log(2048) == int(np.floor(np.log(2.0) * 1024)) and lib.orc_q_log(-5) == 0


@pytest.mark.parametrize("model", ["GCN", "GAT", "PNA", "DGN"])
def test_deterministic_and_thread_independent(oracle, model):
    b = gp.synth_hep10k_batch(10, seed=2, with_eigen=(model == "DGN")) if model in ("PNA", "DGN") else gp.synth_molhiv_batch(40, seed=2)
    w = weights.SYNTH[model](seed=7)
    a, aq = oracle.q_forward(model, b, [w], nthreads=1)
    c, cq = oracle.q_forward(model, b, [w], nthreads=8)
    assert np.array_equal(aq, cq) and np.array_equal(a, c)
    scale = 8192.0 if model == "DGN" else 1024.0
    assert np.array_equal(a, aq.astype(np.float32) / np.float32(scale))
    # order-free sums: a permutation of the edge list of every graph leaves every pattern unchanged
    rng = np.random.default_rng(5)
    eo = b.edge_offsets()
    el, ea = b.edge_list.copy(), b.edge_attr.copy()
    for g in range(b.num_graphs):
        p = rng.permutation(int(eo[g + 1] - eo[g])) + int(eo[g])
        el[eo[g]:eo[g + 1]] = b.edge_list[p]
        ea[eo[g]:eo[g + 1]] = b.edge_attr[p]
    b2 = gp.GraphBatch(b.nums_of_nodes, b.nums_of_edges, b.node_feature, el, ea, b.node_eigen)
    assert np.array_equal(oracle.q_forward(model, b2, [w])[1], aq)


def test_gat_tracks_float_when_nothing_wraps(oracle):
    """GAT's activations stay far inside [-32, 32) for features below 32 (no R8 wrap): Q6.10 then differs from the float oracle by
    accumulated truncation only."""
    b = gp.synth_molhiv_batch(64, seed=3)
    b.node_feature[:, 0] %= 20
    w = weights.synth_gat_weights(seed=7)
    q = oracle.q_forward("GAT", b, [w], nthreads=8)[0]
    f = oracle.gat_forward(b, [w], nthreads=8)
    assert np.abs(q - f).max() < 0.1, np.abs(q - f).max()


@pytest.mark.skipif(not os.path.isdir(REF + "/GCN"), reason="reference weights only exist in the build container")
def test_magnitudes_with_shipped_weights(oracle):
    loaders = {"GCN": weights.load_gcn_weights, "GAT": weights.load_gat_weights, "PNA": weights.load_pna_weights, "DGN": weights.load_dgn_weights}
    lo_hi = {"GCN": (-4.5, 1.5), "GAT": (-2.0, 0.5), "PNA": (-2.5, -0.5), "DGN": (-0.6, -0.2)}
    for model, ld in loaders.items():
        w = ld(f"{REF}/{model}")
        b = gp.synth_molhiv_batch(64, seed=4)
        if model == "DGN":
            b = with_eigen(b, 1)
        q = oracle.q_forward(model, b, [w], nthreads=8)[0]
        lo, hi = lo_hi[model]
        assert np.isfinite(q).all() and lo < float(np.median(q)) < hi, (model, float(np.median(q)))
