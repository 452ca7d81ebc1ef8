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
            name = r["Kernel_Name"]
            # the Gram kernels are launched for lockstep groups of different sizes in one bench.py run (the 8-chain headline group, the 4- and 2-chain per-rank workloads of
            # `expected_scaling`): one row per launch size, so that the headline launch's average can be read off (round 6)
            if "k_gram" in name and "reduce" not in name:
                g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
                w = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)
                if g and w:
                    name = "[%d wgs] %s" % (g // w, name.replace("void ", ""))
            out.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
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
    if any(k.startswith("[") for k in agg):
        print("(rows marked [N wgs]: the Gram kernels by launch size -- a bench.py run times the headline group of 8 chains, then one chain and the 4- and 2-chain per-rank workloads of "
              "`expected_scaling`; the other <bnr_many> rows average over those group sizes: headline-only figures are in the sweep statistics / DESIGN.md section 4)")

if __name__ == "__main__":
    main(sys.argv[1])
