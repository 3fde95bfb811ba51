#!/usr/bin/env python3
"""profiles/limits.json from the committed SQ counter passes (profiles/<tag>_<M>_pmc_SQ.txt and _SQ2.txt, run_round.sh).

  python profiles/make_limits.py r03

What a kernel that keeps its tensors on chip is limited by cannot be read off HBM bytes.  Per dominant kernel this records, as
shares of what the hardware offers over the kernel's duration (GRBM_GUI_ACTIVE / 8 XCDs = clocks of the launch):
  mfma_busy    SQ_VALU_MFMA_BUSY_CYCLES / (clocks x 256 CUs x 4 SIMDs)      matrix pipe
  lds_busy     SQ_LDS_IDX_ACTIVE        / (clocks x 256 CUs)                 LDS array (bank_conflict_share = conflict cycles / these)
  valu_issue   4 x SQ_ACTIVE_INST_VALU  / (clocks x 256 CUs x 4 SIMDs)      VALU issue (the counter is in quad-cycles)
  lds_issue    4 x SQ_ACTIVE_INST_LDS   / (clocks x 256 CUs x 4 SIMDs)      LDS instruction issue
and, as shares of the waves' own cycles (SQ_WAVE_CYCLES): wait_share (SQ_WAIT_ANY: parked on s_waitcnt / barriers) and
issue_stall_share (SQ_WAIT_INST_ANY).  `limit` names the largest of the four resource shares -- what the kernel would hit first if
its waves never waited; `wait_share` says how far the waves are from hitting anything.
`simd_issue` = mfma_busy + valu_issue: the matrix pipe and the VALU of a SIMD do NOT run side by side -- tools/coissue4.hip
(profiles/r05_coissue4.txt): four waves per SIMD, 17.5 clocks per MFMA alone, 3.8-4.4 per VALU instruction alone, and the two
streams together take 0.76-0.87 of the SUM of their times whichever waves carry them (1.04 for v_pk_fma_f32), so the two shares
can add up to 1.15-1.3 at the very most (`simd_issue_ceiling` 1.2).  A kernel at 0.86 is within 30 % of that roof whatever its
distance from the f16 peak says.
bench.py attaches the entry of the kernel it reports as roofline.issue_limit.
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
KERNELS = {
    "GIN": {"gin_resident": "gin_resident_kernel", "gin_aggregate": "gin_aggregate_tiled_kernel"},
    "GIN-VN": {"gin_resident": "gin_resident_kernel"},
    "GCN": {"gcn_resident": "gcn_resident_kernel"},
    "GAT": {"gat_resident": "gat_resident_kernel"},
    "PNA": {"pna_resident": "pna_resident_kernel", "pna_layer_fused": "pna_layer_fused_kernel"},
    "DGN": {"dgn_resident": "dgn_resident_kernel", "dgn_layer_fused": "dgn_layer_mfma_kernel"},
}


def parse(path):
    out, cur = {}, None
    if not os.path.exists(path):
        return out
    for line in open(path).read().splitlines():
        if line.startswith("#") or not line.strip():
            continue
        if not line.startswith(" "):
            cur = line.strip()
            out[cur] = {}
            continue
        m = re.match(r"\s+(\S+)\s+dispatches=(\d+)\s+avg=([\d,\.]+)", line)
        if m and cur is not None:
            out[cur][m.group(1)] = float(m.group(3).replace(",", ""))
            out[cur]["_dispatches"] = max(out[cur].get("_dispatches", 0), int(m.group(2)))
    return out


def main():
    tag = sys.argv[1]
    res = {"_comment": " ".join(__doc__.strip().split("\n\n", 2)[2].split())}
    for model, kernels in KERNELS.items():
        a = parse(os.path.join(HERE, f"{tag}_{model}_pmc_SQ.txt"))
        b = parse(os.path.join(HERE, f"{tag}_{model}_pmc_SQ2.txt"))
        entry = {}
        for name, pat in kernels.items():
            # the instance launched once per step (a kernel's other instances -- the aggregation probe's one-off input pass -- have fewer dispatches)
            ka = max((v for k, v in a.items() if pat in k), key=lambda v: v.get("_dispatches", 0), default=None)
            kb = max((v for k, v in b.items() if pat in k), key=lambda v: v.get("_dispatches", 0), default=None)
            if not ka:
                continue
            clk = ka["GRBM_GUI_ACTIVE"] / 8.0
            rec = {"mfma_busy": ka["SQ_VALU_MFMA_BUSY_CYCLES"] / (clk * 256 * 4), "lds_busy": ka["SQ_LDS_IDX_ACTIVE"] / (clk * 256),
                   "bank_conflict_share": ka["SQ_LDS_BANK_CONFLICT"] / max(ka["SQ_LDS_IDX_ACTIVE"], 1.0),
                   "wait_share": ka["SQ_WAIT_ANY"] / max(ka["SQ_WAVE_CYCLES"], 1.0)}
            if kb:
                clk2 = kb["GRBM_GUI_ACTIVE"] / 8.0
                rec["valu_issue"] = 4.0 * kb["SQ_ACTIVE_INST_VALU"] / (clk2 * 256 * 4)
                rec["lds_issue"] = 4.0 * kb["SQ_ACTIVE_INST_LDS"] / (clk2 * 256 * 4)
                rec["issue_stall_share"] = kb["SQ_WAIT_INST_ANY"] / max(kb["SQ_WAVE_CYCLES"], 1.0)
                rec["valu_insts_per_launch"] = kb["SQ_INSTS_VALU"]
            if "valu_issue" in rec:
                rec["simd_issue"] = rec["mfma_busy"] + rec["valu_issue"]
                rec["simd_issue_ceiling"] = 1.2
            res_keys = [k for k in ("mfma_busy", "lds_busy", "valu_issue", "lds_issue") if k in rec]
            rec["limit"] = max(res_keys, key=lambda k: rec[k])
            rec = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in rec.items()}
            rec["files"] = f"profiles/{tag}_{model}_pmc_SQ.txt, profiles/{tag}_{model}_pmc_SQ2.txt"
            entry[name] = rec
        if entry:
            res[model] = entry
    json.dump(res, open(os.path.join(HERE, "limits.json"), "w"), indent=2)
    print(json.dumps(res, indent=2))


if __name__ == "__main__":
    main()
