"""Concurrent-engine stress test of every resident / fused kernel (round-4 verdict, item 2).

The kernels that stream weights through LDS (`global_load_lds` -> `s_waitcnt vmcnt(0)` -> `s_barrier` -> `ds_read`) are exactly the
ones a timing-dependent hazard would show in -- and only when other work shares the device: a second and fourth engine of the same
model on their own streams and host threads, plus one engine of ANOTHER model hammering the GPU the whole time.  Every engine
computes the same batch RUNS times; every result must have the bits of the first, quiet, single-engine run.  (On the 8-GPU node the
C-ABI group runs one such engine per device and the entry points two; this is the same concurrency on one device.)"""
import threading

import numpy as np
import pytest

from flowgnn_amd import Engine, graphpack as gp, weights

pytestmark = pytest.mark.gpu

RUNS = 200  # runs of the whole model per engine, at two and at four concurrent engines


def stress_batch(model):
    # several graph tiles per CU so that both engines' persistent workgroups are in flight together
    if model in ("GIN", "GAT"):
        return gp.synth_molhiv_batch(6000, seed=141)
    if model == "GIN-VN":
        return gp.add_virtual_nodes(gp.synth_molhiv_batch(6000, seed=142))
    if model == "GCN":
        return gp.synth_molpcba_batch(6000, seed=143)
    return gp.synth_hep10k_batch(3000, seed=144, with_eigen=(model == "DGN"))


HAMMER = {"GIN": "PNA", "GIN-VN": "GCN", "GCN": "GAT", "GAT": "DGN", "PNA": "GIN", "DGN": "GCN"}


@pytest.mark.parametrize("model", ["GIN", "GIN-VN", "GCN", "GAT", "PNA", "DGN"])
def test_concurrent_engines_keep_their_bits(model):
    b, w = stress_batch(model), weights.SYNTH[model](seed=7)
    quiet = Engine(model, device=0)
    try:
        quiet.set_weights(w)
        want = quiet.forward(b).copy()
        assert np.isfinite(want).all()
        assert np.array_equal(quiet.forward(b), want)  # deterministic when alone, or nothing below means anything
    finally:
        quiet.close()

    hm = HAMMER[model]
    hb, hw = stress_batch(hm), weights.SYNTH[hm](seed=9)
    stop = threading.Event()
    hammer_runs = [0]
    errors = []

    def hammer():
        try:
            e = Engine(hm, device=0)
            try:
                e.set_weights(hw)
                first = e.forward(hb).copy()
                while not stop.is_set():
                    e.run()
                    hammer_runs[0] += 1
                    if hammer_runs[0] % 16 == 0 and not np.array_equal(e.results(), first):
                        errors.append(f"hammer {hm}: run {hammer_runs[0]} differs from its first run")
                e.sync()
            finally:
                e.close()
        except Exception as ex:  # noqa: BLE001 - reported by the main thread
            errors.append(f"hammer {hm}: {ex!r}")

    def worker(idx, n_engines, bad):
        try:
            e = Engine(model, device=0)
            try:
                e.set_weights(w)
                e.set_batch(b)
                for it in range(RUNS):
                    e.run()
                    got = e.results()
                    if not np.array_equal(got, want):
                        d = np.nonzero(got != want)[0]
                        bad.append((idx, it, int(d.size), float(np.abs(got - want).max())))
            finally:
                e.close()
        except Exception as ex:  # noqa: BLE001
            errors.append(f"engine {idx}: {ex!r}")

    ht = threading.Thread(target=hammer)
    ht.start()
    try:
        for n_engines in (2, 4):
            bad = []
            ts = [threading.Thread(target=worker, args=(i, n_engines, bad)) for i in range(n_engines)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            assert not errors, errors
            assert not bad, f"{model} x{n_engines} beside {hm}: {len(bad)} runs differ from the quiet run, first (engine, run, values, max|d|): {bad[:4]}"
    finally:
        stop.set()
        ht.join()
    assert not errors, errors
    assert hammer_runs[0] > 0
