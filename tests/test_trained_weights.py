"""The TRAINED weight sets the reference ships (tests/golden/ref_weights_<model>.npz, made by tests/golden/make_ref_weights.py from the
reference's .bin files) on the GPU: the synthetic sets have the shipped sets' per-tensor scales, but not their structure -- GIN logits of
-3 .. -6, PNA's saturating readout head, BatchNorm statistics of a trained GCN.  CPU: the fixtures are what the loaders read from the
reference (when it is present) and the oracle reproduces the `logits_reference_weights` of the per-model fixtures from them.  GPU: the
engine does, through the C ABI, on the graph-resident / fused kernels and on the per-layer ones; the fixed-point modes stay bit-exact."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_modes_golden as mk  # noqa: E402  (fixture_batch)

MODELS = ["GIN", "GCN", "GAT", "PNA", "DGN"]
FRAC = {"GIN": 10, "GCN": 10, "GAT": 10, "PNA": 10, "DGN": 13}
SWITCH = {"GIN": "FLOWGNN_GIN_RESIDENT", "GCN": "FLOWGNN_GCN_RESIDENT", "GAT": "FLOWGNN_GAT_RESIDENT", "PNA": "FLOWGNN_PNA_FUSED", "DGN": "FLOWGNN_DGN_FUSED"}


def trained(model):
    z = np.load(os.path.join(HERE, "golden", f"ref_weights_{model.lower()}.npz"))
    return {k: z[k] for k in z.files}


def expected(model):
    return np.load(os.path.join(HERE, "golden", mk.FIXTURES[model] + ".npz"))["logits_reference_weights"]


def tol(want):
    return dict(rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(want).max())))


@pytest.mark.parametrize("model", MODELS)
def test_fixture_is_the_shipped_set_and_the_oracle_reproduces_its_logits(model, oracle):
    from flowgnn_amd import weights
    w = trained(model)
    ref = os.path.join(os.environ.get("FLOWGNN_REFERENCE", "/root/reference"), model)
    if os.path.isdir(ref):
        shipped = weights.LOADERS[model](ref)
        assert sorted(shipped) == sorted(w)
        assert all(np.array_equal(np.asarray(shipped[k], np.float32), w[k]) for k in w)
    got = getattr(oracle, model.lower() + "_forward")(mk.fixture_batch(model), [w])
    want = expected(model)
    assert np.allclose(got, want, rtol=1e-6, atol=1e-6), np.abs(got - want).max()
    assert np.isfinite(want).all() and float(np.abs(want).max()) > 0.5  # trained logits, not noise around zero


@pytest.mark.gpu
@pytest.mark.parametrize("per_layer", [False, True], ids=["resident-or-fused", "per-layer"])
@pytest.mark.parametrize("model", MODELS)
def test_gpu_matches_the_oracle_on_trained_weights(model, per_layer, monkeypatch):
    from flowgnn_amd import Engine
    if per_layer:
        monkeypatch.setenv(SWITCH[model], "0")
    e = Engine(model, device=0)
    try:
        e.set_weights(trained(model))
        got = e.forward(mk.fixture_batch(model))
    finally:
        e.close()
    want = expected(model)
    assert np.allclose(got, want, **tol(want)), (model, np.abs(got - want).max(), float(np.abs(want).max()))


@pytest.mark.gpu
def test_gin_vn_on_trained_weights(oracle):
    from flowgnn_amd import Engine, graphpack as gp
    w = trained("GIN")  # GIN-VN ships the same files
    b = gp.add_virtual_nodes(mk.fixture_batch("GIN"))
    want = oracle.gin_forward(b, [w], nthreads=8)
    e = Engine("GIN-VN", device=0)
    try:
        e.set_weights(w)
        got = e.forward(b)
    finally:
        e.close()
    assert np.isfinite(want).all()
    assert np.allclose(got, want, rtol=2e-4, atol=1e-3 * max(1.0, float(np.abs(want).max()))), np.abs(got - want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("model", MODELS)
def test_fixed_point_mode_bit_exact_on_trained_weights(model, oracle):
    from flowgnn_amd import Engine
    w, b = trained(model), mk.fixture_batch(model)
    if model == "GIN":
        _, want_q = oracle.gin_forward_q(b, [w], nthreads=8)
    else:
        _, want_q = oracle.q_forward(model, b, [w], nthreads=8)
    e = Engine(model, device=0)
    try:
        e.set_weights(w)
        e.set_numeric_mode("q6.10")
        got = e.forward(b)
    finally:
        e.close()
    assert np.array_equal(np.round(got * float(1 << FRAC[model])).astype(np.int64), want_q.astype(np.int64))
