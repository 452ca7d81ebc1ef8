#!/usr/bin/env python3
"""Durations of consecutive factorization launches in a rocprofv3 kernel trace: tools/chol_seq.py <trace.csv> [kernel substring] [skip]"""
import csv, sys
path, sub, skip = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'k_chol_ll<bnr_many>'), int(sys.argv[3]) if len(sys.argv) > 3 else 1600
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Grid_Size_Y']))
rows.sort()
seq = [r for r in rows if sub in r[2]][skip:skip + 32]
print(' '.join('%.1f' % ((r[1] - r[0]) / 1e3) for r in seq))
print('gaps', ' '.join('%.1f' % ((b[0] - a[1]) / 1e3) for a, b in zip(seq, seq[1:])))
