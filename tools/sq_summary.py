#!/usr/bin/env python3
"""Per-kernel means of the SQ counters of one rocprofv3 pass (`--kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS`, tools/profile_round.sh).
usage: sq_summary.py <dir>/run_counter_collection.csv > sq_counters.json"""
import csv, json, sys
from collections import defaultdict


def main():
    tot = defaultdict(lambda: defaultdict(float))
    n = defaultdict(lambda: defaultdict(int))
    with open(sys.argv[1], newline="") as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            if "gnn::" not in k:
                continue
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            n[k][row["Counter_Name"]] += 1
    out = {"note": "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT "
                   "SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS over `bench.py --steps 400 --warmup 100 --no-cpu-baseline` (one pass, no "
                   "other trace domain); means per launch", "kernels": {}}
    for k in tot:
        d = {"launches": max(n[k].values())}
        for c in sorted(tot[k]):
            d[c] = round(tot[k][c] / n[k][c])
        wc = d.get("SQ_WAVE_CYCLES") or 1
        d["wait_any_frac"] = round(d.get("SQ_WAIT_ANY", 0) / wc, 3)
        d["active_inst_frac"] = round(d.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
        d["wait_inst_frac"] = round(d.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
        if d.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict_frac_of_lds_active"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"], 3)
        out["kernels"][k[:100]] = d
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
