#!/usr/bin/env python3
"""Per-sweep statistics of a rocprofv3 --kernel-trace run of bench.py (8-chain group, graph-replayed region): when the scalar branch starts and ends relative to
the Gram's start, how often it ends after the factorization, and what such a sweep costs.   usage: sweep_stats.py <dir> [first sweep] [last sweep]"""
import csv, glob, os, statistics, sys, collections

def main(d, lo=30, hi=480):
    rows = []
    for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(p) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    g = [i for i, r in enumerate(rows) if "k_gram8<bnr_many" in r[2] or "k_gram_i8<bnr_many" in r[2]]
    out = []
    for a, b in zip(g, g[1:]):
        seg = rows[a:b]
        t0 = seg[0][0]
        tail = [r for r in seg if "k_tail<bnr_many" in r[2]]
        rhs = [r for r in seg if "k_rhs<bnr_many" in r[2]]
        ch = [r for r in seg if "k_chol_step" in r[2]]
        sw = [r for r in seg if "k_solve_w" in r[2]]
        if not tail or not rhs or len(ch) != 16 or not sw:
            continue
        out.append(((tail[0][0] - t0) / 1e3, (rhs[0][1] - t0) / 1e3, (ch[-1][1] - t0) / 1e3, (sw[0][0] - t0) / 1e3, (rows[b][0] - t0) / 1e3, (ch[-1][1] - ch[0][0]) / 1e3))
    gr = out[lo:hi]
    ts = sorted(o[0] for o in gr)
    late = [o for o in gr if o[1] > o[2]]
    good = [o for o in gr if o[1] <= o[2]]
    print("%d sweeps; k_tail starts (us after the Gram): min %.0f median %.0f p90 %.0f max %.0f" % (len(gr), ts[0], statistics.median(ts), ts[int(.9 * len(ts))], ts[-1]))
    print("histogram of the start (20 us bins):", sorted(collections.Counter(int(t // 20) * 20 for t in ts).items()))
    print("scalar branch ends after the factorization in %d of %d sweeps" % (len(late), len(gr)))
    print("sweep length: mean %.1f median %.1f | those sweeps %.1f | the others %.1f" % (statistics.mean(o[4] for o in gr), statistics.median(o[4] for o in gr),
          statistics.mean(o[4] for o in late) if late else 0.0, statistics.mean(o[4] for o in good) if good else 0.0))
    print("factorization (16 launches, first start to last end): mean %.1f; k_solve_w starts %.1f us after it on average" % (statistics.mean(o[5] for o in gr), statistics.mean(o[3] - o[2] for o in gr)))

if __name__ == "__main__":
    main(sys.argv[1], *(int(x) for x in sys.argv[2:4]))
