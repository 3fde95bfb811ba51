"""The drop-in symbol with HOST arrays (GIN, 2^18 molhiv graphs = 536 MB of int32): option h2d_pack (host threads that narrow the arrays
for the transfer; 0 = plain copies) x engines on the one device (the entry points' default: two) x ranges per engine."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from flowgnn_amd import compute_graphs, graphpack as gp, weights, entry_set_devices, entry_set_option, entry_set_pipeline
b = gp.synth_molhiv_batch(1 << 18, seed=1234)
w = weights.synth_gin_weights(seed=7)
outs = {}
for rep in range(2):
    for engines in (2, 3, 4):
        entry_set_devices([0] * engines)
        for chunks in (0, 3):
            entry_set_pipeline(chunks)
            for pack in (0, 16):
                entry_set_option("GIN", "h2d_pack", pack)
                compute_graphs("GIN", b, [w])
                ts = []
                for _ in range(4):
                    t0 = time.perf_counter(); out = compute_graphs("GIN", b, [w]); ts.append(time.perf_counter() - t0)
                outs[(engines, chunks, pack)] = out
                print(f"engines={engines} chunks={chunks} h2d_pack={pack:2d}: {min(ts)*1e3:.2f} ms  ({b.num_graphs/min(ts)/1e6:.1f} M graphs/s)", flush=True)
ref = next(iter(outs.values()))
print("same bits:", all(np.array_equal(ref, v) for v in outs.values()))
