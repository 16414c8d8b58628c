#!/usr/bin/env python3
"""One step's dispatches in order, from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`):
name, grid, duration and the gap to the previous dispatch's end.  Usage: tools/step_trace.py TRACE.csv [N [SKIP]]
prints the LAST N dispatches (default 16) -- the tail of the timed loop, past warm-up -- after dropping the last SKIP."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(.*\)$", "", name).replace("gnn::", "").replace("void ", "")
    return name[:100]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if skip:
        rows = rows[:-skip]
    rows = rows[-n:]
    prev = None
    tot = 0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev) / 1e3 if prev else 0.0
        tot += e - s
        print("%8.2f us  gap %6.2f  grid %7s wg %4s  lds %6s  %s" % ((e - s) / 1e3, gap, r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"),
                                                          r.get("LDS_Block_Size", "?"), short(r["Kernel_Name"])))
        prev = e
    print("sum of durations %.2f us; first start -> last end %.2f us" % (tot / 1e3, (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3))


if __name__ == "__main__":
    main()
