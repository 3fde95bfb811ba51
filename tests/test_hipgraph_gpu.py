"""hipGraph replay of flowgnn_run's launch sequence (include/flowgnn.h: flowgnn_graph_replays): replays give the
bytes a plain run gives, for every model, and every call that changes what the kernels read or write drops the recording."""
import numpy as np
import pytest

from flowgnn_amd import Engine, graphpack as gp, weights

pytestmark = pytest.mark.gpu


def small_batch(model, seed):
    if model in ("PNA", "DGN"):
        return gp.synth_hep10k_batch(20, seed=seed, with_eigen=(model == "DGN"))
    b = gp.synth_molpcba_batch(64, seed=seed) if model == "GCN" else gp.synth_molhiv_batch(64, seed=seed)
    return gp.add_virtual_nodes(b) if model == "GIN-VN" else b


@pytest.mark.parametrize("model", ["GIN", "GIN-VN", "GCN", "GAT", "PNA", "DGN"])
def test_replays_are_bit_identical_and_invalidate(model, monkeypatch):
    base = model.replace("-VN", "").lower()
    w1 = getattr(weights, f"synth_{base}_weights")(seed=7)
    w2 = getattr(weights, f"synth_{base}_weights")(seed=8)
    b1, b2 = small_batch(model, 3), small_batch(model, 4)

    monkeypatch.setenv("FLOWGNN_HIPGRAPH", "0")
    plain = Engine(model, device=0)
    monkeypatch.setenv("FLOWGNN_HIPGRAPH", "1")
    e = Engine(model, device=0)
    try:
        plain.set_weights(w1)
        e.set_weights(w1)
        want11 = plain.forward(b1)
        assert plain.graph_replays() == 0
        e.set_batch(b1)
        outs = []
        for _ in range(4):  # plain, captured + launched, replay, replay
            e.run()
            outs.append(e.results())
        assert e.graph_replays() == 3
        for o in outs:
            assert np.array_equal(o, want11)
        h_replay = e.final_h()  # the tap / folded-readout bookkeeping survives a replay
        plain.run()
        assert np.array_equal(h_replay, plain.final_h())

        e.set_weights(w2)  # new device copies of the weights: the recording must go
        plain.set_weights(w2)
        want21 = plain.forward(b1)
        n0 = e.graph_replays()
        e.run()
        assert e.graph_replays() == n0 and np.array_equal(e.results(), want21)
        e.run(); e.run()
        assert e.graph_replays() == n0 + 2 and np.array_equal(e.results(), want21)

        want22 = plain.forward(b2)  # another batch
        assert np.array_equal(e.forward(b2), want22)
        e.run(); e.run()
        assert np.array_equal(e.results(), want22)
    finally:
        e.close()
        plain.close()


def test_range_fallback_drops_the_recording(monkeypatch):
    monkeypatch.setenv("FLOWGNN_HIPGRAPH", "1")
    w = weights.synth_gin_weights(seed=7)
    big = dict(w)
    big["node_embedding_weight"] = w["node_embedding_weight"] * np.float32(1e5)  # leaves the split-f16 range
    b = gp.synth_molhiv_batch(48, seed=5)
    e = Engine("GIN", device=0)
    try:
        e.set_weights(big)
        e.set_batch(b)
        e.run(); first = e.results()   # plain run trips the range flag -> exact re-run
        assert e.exact_reruns() == 1
        e.run(); e.run(); e.run()      # now on the exact kernels: plain, capture, replay
        assert e.exact_reruns() == 1 and e.graph_replays() >= 1
        assert np.array_equal(e.results(), first)
    finally:
        e.close()
