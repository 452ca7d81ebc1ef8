#!/usr/bin/env python3
"""One sweep of a rocprofv3 --kernel-trace run as a timeline: start / end of every dispatch relative to the end of the previous back-projection.
usage: prof_timeline.py <dir> [sweep index from the end, default 40] [kernel that ends a sweep, default k_backproj<bnr_many]"""
import csv, glob, os, re, sys

def main(d, back=40, endk="k_backproj<bnr_many"):
    rows = []
    for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(p) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
    rows.sort()
    bp = [i for i, r in enumerate(rows) if endk in r[2]]
    i0, i1 = bp[-back - 1], bp[-back]
    t0 = rows[i0][1]
    for s, e, name, g in rows[i0 + 1:i1 + 1]:
        short = re.sub(r"^void ", "", name)
        short = re.sub(r"\(.*$", "", short)
        print("%8.1f %8.1f %7.1f  %-40s grid %d" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, short[:40], g))

if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40, sys.argv[3] if len(sys.argv) > 3 else "k_backproj<bnr_many")
