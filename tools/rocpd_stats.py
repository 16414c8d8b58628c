#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`
writes DIR/NAME_results.db on this image): calls, total / average / min / max duration in ns, share of the
GPU time -- the columns of rocprofv3's kernel_stats.csv.  Usage: tools/rocpd_stats.py DB [--csv OUT] [--last N]
(--last N: only the last N dispatches of each kernel, to leave warm-up and set-up launches out)."""
import collections
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*\)$", "", name)
    return name if len(name) < 150 else name[:147] + "..."


def main():
    args = sys.argv[1:]
    out = None
    last = None
    if "--csv" in args:
        i = args.index("--csv"); out = args[i + 1]; del args[i:i + 2]
    if "--last" in args:
        i = args.index("--last"); last = int(args[i + 1]); del args[i:i + 2]
    c = sqlite3.connect(args[0])
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    namecol = "kernel_name" if "kernel_name" in cols else "display_name"
    names = dict(c.execute("select id, %s from %s" % (namecol, ks)))
    per = collections.defaultdict(list)
    for kid, start, end, gx, wx in c.execute("select kernel_id, start, end, grid_size_x, workgroup_size_x from %s order by start" % kd):
        per[names.get(kid, str(kid))].append(end - start)
    rows = []
    for name, d in per.items():
        if last:
            d = d[-last:]
        rows.append((short(name), len(d), sum(d), sum(d) / len(d), min(d), max(d)))
    total = sum(r[2] for r in rows) or 1
    rows.sort(key=lambda r: -r[2])
    lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
    for n, calls, tot, avg, mn, mx in rows:
        lines.append('"%s",%d,%d,%.1f,%.2f,%d,%d' % (n, calls, tot, avg, 100.0 * tot / total, mn, mx))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
