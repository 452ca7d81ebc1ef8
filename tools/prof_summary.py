#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db or *_kernel_trace.csv) per kernel."""
import csv, glob, os, sqlite3, sys

def from_db(path):
    cur = sqlite3.connect(path).cursor()
    return [(r[0], r[1] - 0, r[2]) for r in cur.execute("select name, start, end from kernels")]

def from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    return out

def main(d):
    rows = []
    for p in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        rows += from_db(p)
    for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += from_csv(p)
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0, 10**18, 0])
        a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
    tot = sum(a[1] for a in agg.values()) or 1
    print("%-52s %8s %12s %10s %10s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-52s %8d %12.2f %10.2f %10.2f %6.1f%%" % (name[:52], a[0], a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / tot))
    print("total kernel time: %.3f ms over %d dispatches" % (tot / 1e6, sum(a[0] for a in agg.values())))

if __name__ == "__main__":
    main(sys.argv[1])
