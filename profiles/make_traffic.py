#!/usr/bin/env python3
"""profiles/traffic.json from the committed PMC summaries (one FETCH_SIZE pass and one WRITE_SIZE pass per model).

  python profiles/make_traffic.py r02

bytes per launch = 2 x FETCH_SIZE[KB] x 1000 + WRITE_SIZE[KiB] x 1024:
  * FETCH_SIZE counts the 128-B requests of wide coalesced reads at 64 B on gfx950 (MI355X_MICROARCH.md, HBM) -> doubled;
  * WRITE_SIZE is calibrated on atom_encoder_kernel, which writes exactly N x 400 B (2 620 292.6 KiB for the GIN batch).
bench.py copies the entry of the kernel it reports into roofline.traffic when it runs the same batch size.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
# bench.py kernel name -> substring of the rocprofv3 kernel name (first match wins; the non-final layer variant is listed first)
KERNELS = {
    "GIN": (1 << 18, "GIN", {"gin_resident": "gin_resident_kernel", "gin_layer_fused": "gin_layer_split_kernel",
                             "gin_aggregate": "gin_aggregate_tiled_kernel"}),
    "GCN": (1 << 18, "GCN", {"gcn_resident": "gcn_resident_kernel", "gcn_layer_fused": "gcn_layer_fused_kernel<false>", "gcn_aggregate": "tiled_aggregate_kernel<fg::GcnAggPolicy"}),
    "GAT": (1 << 18, "GAT", {"gat_resident": "gat_resident_kernel", "gat_layer": "gat_layer_kernel<false, false"}),
    "PNA": (1 << 16, "PNA", {"pna_resident": "pna_resident_kernel", "pna_layer_fused": "pna_layer_fused_kernel", "pna_aggregate": "tiled_aggregate_kernel<fg::PnaAggPolicy",
                             "pna_dense": "pna_dense_split_kernel"}),
    "DGN": (1 << 16, "DGN", {"dgn_resident": "dgn_resident_kernel", "dgn_layer_fused": "dgn_layer_mfma_kernel", "dgn_aggregate": "tiled_aggregate_kernel<fg::DgnAggPolicy",
                             "dgn_dense": "dense200_res_relu_split_kernel"}),
}

def table(path):
    out = []
    for line in open(path).read().splitlines()[1:]:
        parts = line.rsplit(None, 3)
        if len(parts) == 4:
            out.append((parts[0], float(parts[2]), int(parts[1])))
    return out


def main():
    tag = sys.argv[1]
    res = {"_comment": __doc__.strip().split("\n\n", 1)[1].replace("\n", " ")}
    for model, (graphs, stem, kernels) in KERNELS.items():
        f = f"{tag}_{stem}_pmc_FETCH_SIZE.txt"
        w = f"{tag}_{stem}_pmc_WRITE_SIZE.txt"
        ft, wt = table(os.path.join(HERE, f)), table(os.path.join(HERE, w))
        entry = {"graphs": graphs}
        for name, pat in kernels.items():
            # the instance launched once per step: a kernel's other instances (the aggregation probe's one-off input pass) have fewer dispatches
            fk = max(((n, v) for k, v, n in ft if pat in k), default=(0, None))[1]
            wk = max(((n, v) for k, v, n in wt if pat in k), default=(0, None))[1]
            if fk is None or wk is None:  # a kernel of an older or a switched-off path: not launched in this round's default run
                continue
            entry[name] = {"bytes": int(round(2 * fk * 1000 + wk * 1024, -6)), "fetch_kb": fk, "write_kb": wk,
                           "source": f"profiles/{f}, profiles/{w}"}
        res[model] = entry
    json.dump(res, open(os.path.join(HERE, "traffic.json"), "w"), indent=2)
    print(json.dumps(res, indent=2))


if __name__ == "__main__":
    main()
