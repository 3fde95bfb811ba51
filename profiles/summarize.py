#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small per-kernel summaries committed under profiles/.

  python profiles/summarize.py stats  <kernel_stats.csv | kernel_trace.csv> [steps warmup]  > profiles/rNN_*.txt
      with `steps warmup` (the bench.py arguments of the traced run) the warm-up launches of every kernel are left out, so that
      the averages are those of the timed region (a kernel with c calls runs c / (steps + warmup) times per step; its first
      warmup x that many calls, in time order, are dropped)
  python profiles/summarize.py pmc    <counter_collection.csv> COUNTER       > profiles/rNN_*.txt
  python profiles/summarize.py pmcall <counter_collection.csv>               > profiles/rNN_*_pmc_SQ.txt   (all counters of a pass)
"""
import csv
import sys
from collections import defaultdict


def stats(path, steps=0, warmup=0):
    rows = list(csv.DictReader(open(path)))
    if rows and "Start_Timestamp" in rows[0]:  # kernel_trace.csv -> aggregate ourselves
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        calls = defaultdict(int)
        for r in rows:
            calls[r["Kernel_Name"]] += 1
        skip = {}
        for k, c in calls.items():
            per_step = c // (steps + warmup) if steps + warmup > 0 and c % (steps + warmup) == 0 else 0
            skip[k] = per_step * warmup  # kernels that do not run once per step (set-up launches) are kept whole ...
            if per_step == 0 and c >= 3 and steps:
                skip[k] = 1  # ... except the first launch of a probe loop (flowgnn_run_aggregation_only: one warm-up launch, cold caches,
                #              then `iters` timed ones -- the figure bench.py reports as `aggregation` is the average of those)
        if steps:
            print(f"# timed region only: {steps} steps, the launches of the {warmup} warm-up step(s) are excluded; of a kernel launched in a "
                  f"probe loop outside the steps (>= 3 launches, not once per step) the first, warm-up launch is excluded")
        agg = defaultdict(lambda: [0, 0.0])
        seen = defaultdict(int)
        for r in rows:
            k = r["Kernel_Name"]
            seen[k] += 1
            if seen[k] <= skip[k]:
                continue
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            a = agg[k]
            a[0] += 1
            a[1] += d
        tot = sum(v[1] for v in agg.values())
        print(f"{'kernel':<70} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'pct':>6}")
        for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"{k[:70]:<70} {c:>7} {t/1e6:>10.3f} {t/c/1e3:>10.2f} {100*t/tot:>6.2f}")
    else:
        cols = rows[0].keys() if rows else []
        print(",".join(cols))
        for r in rows:
            print(",".join(str(r[c])[:80] for c in cols))


def pmc(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    print(f"{'kernel':<70} {'dispatches':>10} {counter + '_avg':>16} {counter + '_total':>18}")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[:70]:<70} {c:>10} {t/c:>16.1f} {t:>18.1f}")


def pmcall(path):
    """every counter of a multi-counter pass, per kernel (averages per dispatch) + the kernel's average duration"""
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    rows = list(csv.DictReader(open(path)))
    print("# columns:", ",".join(rows[0].keys()) if rows else "")
    for r in rows:
        a = agg[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        if "Start_Timestamp" in r and "End_Timestamp" in r:
            d = agg[r["Kernel_Name"]]["(duration_ns)"]
            d[0] += 1
            d[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for k, cs in agg.items():
        print(k[:90])
        for c, (n, t) in sorted(cs.items()):
            print(f"    {c:<34} dispatches={n:<5} avg={t/n:,.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], *(int(a) for a in sys.argv[3:5]))
    elif sys.argv[1] == "pmcall":
        pmcall(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3])
