#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 kernel trace csv: per sweep (k_backproj to k_backproj) wall time, busy time, gaps."""
import csv, sys, glob, os
rows = []
for p in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Stream_Id", "")))
rows.sort()
ends = [i for i, r in enumerate(rows) if r[2] == "k_backproj"]
print("sweeps:", len(ends))
def show(lo, hi):
    t0 = rows[lo][0]
    for s, e, n, st in rows[lo:hi + 1]:
        print("  %9.2f -> %9.2f  (%7.2f us)  %-16s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n, st))
for which in (len(ends) // 4, len(ends) - 30):
    a, b = ends[which] + 1, ends[which + 1]
    print("sweep", which, "wall %.1f us" % ((rows[b][1] - rows[a][0]) / 1e3))
    show(a, b)
per = [(rows[ends[i + 1]][1] - rows[ends[i]][1]) / 1e3 for i in range(len(ends) - 1)]
import statistics
print("median sweep period first half %.1f us, second half %.1f us" % (statistics.median(per[:len(per) // 2]), statistics.median(per[len(per) // 2:])))
