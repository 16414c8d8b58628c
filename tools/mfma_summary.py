#!/usr/bin/env python3
"""Per-kernel MFMA-pipe utilisation from a rocprofv3 counter pass.

usage: mfma_summary.py <dir>/run_counter_collection.csv [more.csv ...] > mfma_counters.json

The pass is `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32
SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py ...` (its own run, no other trace
domains).  Per kernel, averaged over its launches:
  mfma_util      = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)
                   -- the busy cycles summed over the chip's 1024 SIMDs over the cycles the dispatch was active;
                   rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back), and
                   the quotient reads high on dispatches shorter than ~0.3 ms (all of these): it is the share of the
                   kernel's ACTIVE cycles, not of wall time.  This is rocprofv3's own MfmaUtil expression
                   (reduce(SQ_VALU_MFMA_BUSY_CYCLES,sum)/(reduce(GRBM_GUI_ACTIVE,max)*SIMD_NUM)) with max ~ sum/8.
  mfma_flop      = (MOPS_F32 + MOPS_BF16) * 512  -- the FLOPs the MFMA pipe really executed (padding included)
"""
import csv
import json
import sys
from collections import defaultdict

COUNTERS = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE")


def main():
    tot = defaultdict(lambda: defaultdict(float))
    n = defaultdict(lambda: defaultdict(int))
    for path in sys.argv[1:]:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                c = row["Counter_Name"]
                if c in COUNTERS:
                    tot[row["Kernel_Name"]][c] += float(row["Counter_Value"])
                    n[row["Kernel_Name"]][c] += 1
    out = {"note": __doc__.strip().split("\n\n", 1)[1], "kernels": {}}
    for k in sorted(tot):
        if "gnn::" not in k:
            continue
        avg = {c: tot[k][c] / n[k][c] for c in tot[k] if n[k][c]}
        gui = avg.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        busy = avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        out["kernels"][k] = {"launches": max(n[k].values()),
                             "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE_per_xcd": gui,
                             "SQ_BUSY_CYCLES": avg.get("SQ_BUSY_CYCLES"),
                             "mfma_flop": (avg.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) + avg.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)) * 512,
                             "mfma_util": round(busy / (gui * 1024.0), 5) if gui > 0 else None}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
