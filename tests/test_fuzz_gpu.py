"""A short run of the differential fuzzer (scripts/dev/fuzz.py) per model: random batches -- graph sizes 1..500, no edges to
16 in-edges per node, duplicate edges, self loops, isolated nodes, tiny graphs next to 500-node ones -- through the HIP path, against
the CPU oracle, under a batch split and through the drop-in entry point.  (The long runs live outside the test suite; the first one
found the single-node graph with 18 copies of a self loop that test_dgn_rows_whose_in_edges_all_have_zero_weight pins.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from flowgnn_amd import Engine, graphpack as gp, weights

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["f32", "q", "variants"])
@pytest.mark.parametrize("model", ["GIN", "GIN-VN", "GCN", "GAT", "PNA", "DGN"])
def test_short_fuzz(model, mode):
    """f32: against the float oracle (tolerance), a batch split and the entry point; q: the fixed-point mode, BIT-exact against the Q
    oracle and under a split; variants: a random option set (per-layer kernels, unfused paths, ping-pong, ...) against the default."""
    if mode == "q" and model == "GIN-VN":
        pytest.skip("the GIN fixed-point oracle has no virtual-node entry of its own (GIN-VN runs as GIN on augmented graphs)")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dev", "fuzz.py"), model, "5" if mode == "f32" else "3", "7", mode],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "random batches ok" in p.stdout


def test_dgn_rows_whose_in_edges_all_have_zero_weight(oracle):
    """DGN's directional aggregate divides by sum |eig[u] - eig[v]|, with 1 / 8192 in place of 0: a row whose in-edges all come from
    nodes with its own eigenvector entry (self loops; here 18 copies of one) has m2 = 0 EXACTLY in the reference.  The matrix-pipe
    path forms m2 as P - eig_v m1 from two separately rounded sums and must not let 8192 amplify their difference."""
    rng = np.random.default_rng(5)
    nf = np.stack([rng.integers(0, c, 1) for c in (119, 4, 12, 12, 10, 6, 6, 2, 2)], 1).astype(np.int32)
    loops = gp.GraphBatch(np.array([1], np.int32), np.array([18], np.int32), nf, np.zeros((18, 2), np.int32), np.zeros((18, 3), np.int32),
                          np.array([[0, 0.7, 0, 0]], np.float32))
    hep = gp.synth_hep10k_batch(6, seed=3)
    eq = gp.synth_hep10k_batch(2, seed=4)
    eq.node_eigen[:, 1] = 0.25  # every difference is exactly zero in these two graphs
    b = gp.concat_batches([hep.slice(0, 3), loops, eq, hep.slice(3, 6), loops])
    w = weights.SYNTH["DGN"](seed=7)
    want, hd = oracle.dgn_forward(b, [w], dump_h=True, nthreads=4)
    scale = max(1.0, float(np.abs(hd).max()))
    for opts in ({"dgn_mfma_agg": 1}, {"dgn_mfma_agg": 1, "dgn_resident": 0}, {"dgn_mfma_agg": 1, "dgn_resident": 0, "dgn_rowinfo_direct": 0}, {"dgn_mfma_agg": 0}):
        e = Engine("DGN", device=0, options=opts)
        e.set_weights(w)
        got = e.forward(b)
        e.close()
        assert np.allclose(got, want, rtol=2e-4, atol=2e-5 * scale), (opts, np.abs(got - want).max(), scale)
