"""Two engines on one device, two host threads, ranges of one job alternating: where does the time go?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flowgnn_amd import Engine, graphpack as gp, weights
from flowgnn_amd.engine import shard_ranges_c
w = weights.synth_gin_weights(7)
b = gp.synth_molhiv_batch(1 << 18, seed=1234)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NE = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rng = shard_ranges_c(b.nums_of_nodes, b.nums_of_edges, S)
MODE = sys.argv[3] if len(sys.argv) > 3 else "copy"   # copy: every range its own arrays | view: ranges are views of the job's arrays | lock: views + one copier at a time
def view(a, c):
    no, eo = b.node_offsets(), b.edge_offsets()
    return gp.GraphBatch(b.nums_of_nodes[a:c], b.nums_of_edges[a:c], b.node_feature[no[a]:no[c]], b.edge_list[eo[a]:eo[c]], b.edge_attr[eo[a]:eo[c]], None)
parts = [b.slice(a, c) if MODE == "copy" else view(a, c) for a, c in rng]
LOCK = threading.Lock()
engs = [Engine("GIN", 0) for _ in range(NE)]
for e in engs:
    e.set_weights(w)
    for p in parts:
        e.set_batch(p); e.run(); e.sync()
log = []
def work(i):
    e = engs[i]
    for j in range(i, S, NE):
        t0 = time.perf_counter()
        if MODE == "lock":
            with LOCK:
                e.set_batch(parts[j])
        else:
            e.set_batch(parts[j])
        t1 = time.perf_counter(); e.run(); t2 = time.perf_counter(); r = e.results(); t3 = time.perf_counter()
        log.append((i, j, t0, t1, t2, t3))
for rep in range(2):
    log.clear()
    T0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(NE)]
    [t.start() for t in th]; [t.join() for t in th]
    T1 = time.perf_counter()
print(f"total {1e3 * (T1 - T0):.2f} ms")
for i, j, t0, t1, t2, t3 in sorted(log, key=lambda x: x[2]):
    print(f"  eng {i} range {j}: start {1e3 * (t0 - T0):6.2f}  set_batch {1e3 * (t1 - t0):5.2f}  run {1e3 * (t2 - t1):5.2f}  results {1e3 * (t3 - t2):5.2f}")
