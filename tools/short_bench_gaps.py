#!/usr/bin/env python3
"""Where a short timed region loses time: idle gaps (> 12 us) between consecutive kernels of a rocprofv3 kernel trace of `bench.py --steps 20 --warmup 5`,
around the 20 timed sweeps (the last 20 k_gram8 launches before the eager profiling pass)."""
import csv, sys, glob, os
rows = []
for p in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
grams = [i for i, r in enumerate(rows) if r[2].startswith("k_gram8<bnr_many")]
print("k_gram8 launches:", len(grams))
# group launches into bursts separated by > 1 ms
bursts, cur = [], [grams[0]]
for a, b in zip(grams, grams[1:]):
    if rows[b][0] - rows[a][1] > 1_000_000: bursts.append(cur); cur = []
    cur.append(b)
bursts.append(cur)
for bi, b in enumerate(bursts):
    lo, hi = b[0], b[-1]
    # extend to the kernels around
    while lo > 0 and rows[lo][0] - rows[lo - 1][1] < 200_000: lo -= 1
    while hi + 1 < len(rows) and rows[hi + 1][0] - rows[hi][1] < 200_000: hi += 1
    span = (max(r[1] for r in rows[lo:hi + 1]) - rows[lo][0]) / 1e3
    print("burst %d: %d Grams, span %.1f us = %.1f us per Gram" % (bi, len(b), span, span / len(b)))
    end = rows[lo][1]
    for i in range(lo + 1, hi + 1):
        gap = (rows[i][0] - end) / 1e3
        if gap > 12: print("    idle %.1f us before %s (at +%.1f us)" % (gap, rows[i][2][:30], (rows[i][0] - rows[lo][0]) / 1e3))
        end = max(end, rows[i][1])
