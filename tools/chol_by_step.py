#!/usr/bin/env python3
"""Per-launch-index durations of the factorization kernel from a rocprofv3 kernel trace: tools/chol_by_step.py <dir> <kernel substring> <launches per sweep>"""
import sys, csv, glob, collections
d, name, per = sys.argv[1], sys.argv[2], int(sys.argv[3])
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = [r for r in csv.DictReader(open(f)) if name in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[(len(rows) % per):] if len(rows) % per else rows
rows = rows[per * 3:]                                  # skip the first sweeps
acc = collections.defaultdict(list)
gap = collections.defaultdict(list)
for i, r in enumerate(rows):
    acc[i % per].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if i % per: gap[i % per].append((int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3)
tot = 0.0
for k in range(per):
    m = sum(acc[k]) / len(acc[k]); tot += m
    print("%3d  %7.2f us   gap before %5.2f" % (k, m, sum(gap[k]) / len(gap[k]) if gap[k] else 0.0))
print("sum %.1f us over %d sweeps" % (tot, len(acc[0])))
