"""NUM_TASK as a run-time dimension of the readout (GIN/src/dcl.h:25,80,92-93: out[][NUM_TASK],
graph_pred_weights_in[][NUM_TASK][EMB_DIM]): GIN / GIN-VN / GCN with ogbg-molpcba's 128 tasks and a small odd count, HIP path
through the handle API and through the <M>_compute_graphs entry points vs the C oracle."""
import numpy as np
import pytest

from flowgnn_amd import Engine, FlowGNNError, compute_graphs, graphpack as gp, weights

pytestmark = pytest.mark.gpu


def close(a, b):
    return np.allclose(a, b, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tasks", [128, 3])
@pytest.mark.parametrize("env", [{}, {"gin_resident": 0}, {"gin_mfma": "f32"}], ids=["resident", "per-layer", "f32"])
def test_gin_multi_task(oracle, tasks, env):
    w = weights.synth_gin_weights(seed=7, num_tasks=tasks)
    b = gp.synth_molpcba_batch(300, seed=5)
    e = Engine("GIN", device=0, options=env)
    e.set_num_tasks(tasks)
    e.set_weights(w)
    got = e.forward(b)
    want, hd = oracle.gin_forward(b, [w], num_tasks=tasks, dump_h=True, nthreads=8)
    assert got.shape == (300, tasks) and np.isfinite(got).all()
    assert close(got, want), np.abs(got - want).max()
    assert close(e.final_h(), hd[5])
    # back to a single task on the same engine: weights and batch must be set again, results are 1-D again
    e.set_num_tasks(1)
    with pytest.raises(FlowGNNError):
        e.run()
    w1 = weights.synth_gin_weights(seed=7)
    e.set_weights(w1)
    assert close(e.forward(b), oracle.gin_forward(b, [w1], nthreads=8))
    e.close()


def test_gin_vn_multi_task(oracle):
    w = weights.synth_gin_weights(seed=9, num_tasks=12)
    b = gp.add_virtual_nodes(gp.synth_molhiv_batch(64, seed=6))
    e = Engine("GIN-VN", device=0)
    e.set_num_tasks(12)
    e.set_weights(w)
    got, want = e.forward(b), oracle.gin_forward(b, [w], num_tasks=12, nthreads=8)
    assert np.allclose(got, want, rtol=2e-4, atol=1e-3), np.abs(got - want).max()
    e.close()


@pytest.mark.parametrize("env", [{}, {"gcn_unfused": 1}, {"gcn_mfma": "f32"}], ids=["fused", "unfused", "f32"])
def test_gcn_multi_task(tmp_path, oracle, env):
    tasks = 128
    w = weights.synth_gcn_weights(seed=7, num_tasks=tasks)
    b = gp.synth_molpcba_batch(300, seed=8)
    e = Engine("GCN", device=0, options=env)
    e.set_num_tasks(tasks)
    e.set_weights(w)
    got, want = e.forward(b), oracle.gcn_forward(b, [w], num_tasks=tasks, nthreads=8)
    assert got.shape == (300, tasks) and close(got, want), np.abs(got - want).max()
    # the .all.bin loader with the head at its flattened-state_dict offsets
    weights.save_gcn_weights(w, str(tmp_path))
    e.load_weights_dir(str(tmp_path))
    assert close(e.forward(b), want)
    e.close()


def test_entry_points_and_refusals(oracle):
    tasks = 5
    b = gp.synth_molhiv_batch(9, seed=5)
    w, w2 = weights.synth_gin_weights(seed=7, num_tasks=tasks), weights.synth_gin_weights(seed=8, num_tasks=tasks)
    rw = np.array([1, 0, 0, 0, 1, 0, 0, 0, 0], np.int32)
    got = compute_graphs("GIN", b, [w, w2], rw, num_tasks=tasks)
    want = oracle.gin_forward(b, [w, w2], reload_weights=rw, num_tasks=tasks)
    assert got.shape == (9, tasks) and close(got, want), np.abs(got - want).max()
    # the entry-point engine goes back to NUM_TASK = 1 afterwards
    w1 = weights.synth_gin_weights(seed=7)
    assert close(compute_graphs("GIN", b, [w1]), oracle.gin_forward(b, [w1]))
    # GCN through its _mt entry point as well
    wg = weights.synth_gcn_weights(seed=7, num_tasks=tasks)
    gotg = compute_graphs("GCN", b, [wg], num_tasks=tasks)
    assert gotg.shape == (9, tasks) and close(gotg, oracle.gcn_forward(b, [wg], num_tasks=tasks))
    for model in ("PNA", "DGN", "GAT"):
        with pytest.raises(FlowGNNError) as ei:  # single-task MLP heads: refused, never a half-written [G][T] array
            compute_graphs(model, b, [weights.SYNTH[model](seed=1)], num_tasks=4)
        assert ei.value.code == 8
        e = Engine(model, device=0)
        with pytest.raises(FlowGNNError) as ei:
            e.set_num_tasks(4)
        assert ei.value.code == 8
        e.set_num_tasks(1)
        e.close()
    e = Engine("GIN", device=0)
    e.set_num_tasks(4)
    with pytest.raises(FlowGNNError):  # the Q6.10 readout is single-task, as the reference's
        e.set_numeric_mode("q6.10")
    e.close()
