#!/usr/bin/env python3
"""Per-kernel L2<->fabric bytes per launch from two rocprofv3 counter passes.

usage: pmc_summary.py <fetch_dir>/run_counter_collection.csv <write_dir>/run_counter_collection.csv > pmc_traffic.json

The passes are `rocprofv3 --kernel-trace --pmc FETCH_SIZE ...` and `... --pmc WRITE_SIZE ...` (separate
runs, no other trace domains).  Units are KB; FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM
section) prescribes for gfx950 (128-B fabric reads are counted at 64 B)."""
import csv, json, sys
from collections import defaultdict

def collect(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            tot[row["Kernel_Name"]] += float(row["Counter_Value"])
            n[row["Kernel_Name"]] += 1
    return tot, n

def main():
    fetch, nf = collect(sys.argv[1], "FETCH_SIZE")
    write, nw = collect(sys.argv[2], "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over `bench.py --steps 200 "
                   "--warmup 20 --no-cpu-baseline` (MI355X). Units are KB. Correction per MI355X_MICROARCH.md (HBM): "
                   "FETCH_SIZE counts 128-B fabric read requests at 64 B, so read bytes = 2*FETCH_SIZE*1024; WRITE_SIZE "
                   "is exact for 16-B-per-lane stores. These are L2<->fabric bytes summed over the 8 XCDs (Infinity-Cache "
                   "hits included): each XCD's L2 starts cold every kernel, so panels shared by tiles on different XCDs "
                   "are fetched once per XCD.", "kernels": {}}
    for k in fetch:
        if "gnn::" not in k or k not in write or not any(w in k for w in ("fwd_first", "middle4", "rowblock", "grad_update", "tile_step")):
            continue
        f, w = fetch[k] / nf[k], write[k] / nw[k]
        out["kernels"][k] = {"FETCH_SIZE": f, "launches_FETCH_SIZE": nf[k], "WRITE_SIZE": w, "launches_WRITE_SIZE": nw[k],
                             "traffic_bytes_per_launch": (2 * f + w) * 1024}
    json.dump(out, sys.stdout, indent=1)

if __name__ == "__main__":
    main()
