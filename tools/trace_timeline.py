#!/usr/bin/env python3
"""Print a window of a rocprofv3 kernel trace as a timeline (start, end, duration, kernel, grid) -- who overlaps whom."""
import csv, sys, glob, os
rows = []
for p in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
count = int(sys.argv[3]) if len(sys.argv) > 3 else 200
t0 = rows[skip][0]
for s, e, n, g, qd, st in rows[skip:skip + count]:
    print("%9.2f -> %9.2f (%7.2f us) %-34s grid %-8s q %s st %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n[:34], g, qd, st))
