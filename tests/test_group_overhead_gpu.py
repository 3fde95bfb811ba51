"""Host cost of the multi-device group's call path, measured with device 0 listed eight times (round-4 verdict item 6): every engine
has a persistent host thread (GroupWorkers, engine.hip), so the timed loop of `host --devices` -- flowgnn_group_run back to back --
pays no thread creation.  Measured on MI355X (scripts/dev/group_overhead.py): the run call returns in 17-20 us with eight engines
(6-7 us for one engine), and eight engines sharing ONE GPU finish a 2^18-graph step within 1.5 % of one engine.  The bars below are
loose on purpose (shared CI boxes): they catch a return to thread-per-call (60-100 us per call) or a serialised group, not noise."""
import time

import numpy as np
import pytest

from flowgnn_amd import Engine, EngineGroup, graphpack as gp, weights

pytestmark = pytest.mark.gpu


def timed(obj, steps):
    for _ in range(5):
        obj.run()
    obj.sync()
    t0 = time.perf_counter()
    calls = []
    for _ in range(steps):
        c0 = time.perf_counter()
        obj.run()
        calls.append(time.perf_counter() - c0)
    obj.sync()
    # the MEDIAN call: a shared box preempts the caller or a worker now and then (a mean of 200 calls was 67 us on a box whose
    # median was 20), and thread creation per call would show in every call
    return (time.perf_counter() - t0) / steps * 1e6, float(np.median(calls)) * 1e6


def best_of(obj, steps, tries=3):
    runs = [timed(obj, steps) for _ in range(tries)]
    return min(r[0] for r in runs), min(r[1] for r in runs)


def test_eight_engines_on_one_device_cost_microseconds_of_host_time():
    w = weights.synth_gin_weights(seed=7)
    small, large = gp.synth_molhiv_batch(4113, seed=1234), gp.synth_molhiv_batch(1 << 16, seed=1234)
    res = {}
    for name, b, steps in (("small", small, 200), ("large", large, 40)):
        e = Engine("GIN", 0)
        g = EngineGroup("GIN", [0] * 8)
        try:
            e.set_weights(w); e.set_batch(b)
            g.set_weights(w); g.set_batch(b)
            want = e.forward(b)
            assert np.array_equal(g.forward(b), want)
            res[name] = (best_of(e, steps), best_of(g, steps))
        finally:
            e.close(); g.close()
    (one_s, call1_s), (grp_s, call8_s) = res["small"]
    (one_l, call1_l), (grp_l, call8_l) = res["large"]
    print(f"4113 graphs: one engine {one_s:.0f} us/step (call {call1_s:.1f}), 8 engines {grp_s:.0f} (call {call8_s:.1f}); "
          f"65536 graphs: {one_l:.0f} (call {call1_l:.1f}) vs {grp_l:.0f} (call {call8_l:.1f})")
    assert call8_s < 60.0 and call8_l < 60.0, (call8_s, call8_l)  # measured 17-20 us; a std::thread per engine and call is 60-100
    assert grp_l < 1.25 * one_l, (grp_l, one_l)                    # measured 1.01-1.06 (best of three runs each): the shards run side by side on the one GPU
