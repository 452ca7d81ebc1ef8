#!/usr/bin/env python3
"""Round 4: the data-flow factorization (factor_variant 4) against the launch-per-panel one (0): bitwise equal tables over shapes, time per sweep."""
import sys, os, time, hashlib
os.environ.setdefault("BNR_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesiannetworkregression.jl_amd", "csrc", "_var", "exp.so"))   # tools/r4_build_variants.sh "exp:-DBNR_EXPERIMENTS"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
from oracle import bnr_oracle as bo

def tables(n, V, R, variant, nb, rows=6):
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=7)
    ch = bnr_amd.Chain(X, y, R, rows, 3, 1)
    mates = [bnr_amd.Chain.like(ch, 3, c, rows) for c in range(2, nb + 1)]
    for c in [ch] + mates: c.init_prior()
    g = bnr_amd.Group([ch] + mates) if nb > 1 else ch
    g.set_option("factor_variant", variant)
    g.run(2, rows, rows)
    out = [c.fetch() for c in ([ch] + mates)[:3]], ch.counters()
    if nb > 1: g.close()
    for c in [ch] + mates: c.close()
    return out

bad = 0
if "--time-only" not in sys.argv:
  for (n, V, R) in [(500, 100, 7), (70, 19, 5), (130, 12, 3), (64, 9, 2), (193, 30, 5), (1, 5, 2), (33, 2, 1), (512, 20, 4), (449, 15, 3)]:
    for nb in (1, 3, 8):
        try:
            t0, c0 = tables(n, V, R, 0, nb)
            t1, c1 = tables(n, V, R, 4, nb)
        except Exception as e:
            print("FAILED", (n, V, R), nb, e, flush=True); bad += 1; continue
        for i, (a, b) in enumerate(zip(t0, t1)):
            for k in bo.COLUMNS:
                if not np.array_equal(a[k], b[k], equal_nan=True):
                    print("MISMATCH", (n, V, R), "chains", nb, "member", i, k, float(np.nanmax(np.abs(a[k] - b[k])))); bad += 1
        print((n, V, R), "chains", nb, "ok" if not bad else "", {k: v for k, v in c1.items() if v and k != 'where'}, flush=True)
  print("mismatches:", bad)
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
K, W = 1200, 200
for nb in (1, 2, 4, 8):
    for variant in (0, 4):
        tot = K + W
        ch = bnr_amd.Chain(X, y, 7, tot, 5, 1)
        members = [ch] + [bnr_amd.Chain.like(ch, 5, c, tot) for c in range(2, nb + 1)]
        for c in members: c.init_prior()
        g = bnr_amd.Group(members) if nb > 1 else ch
        g.set_option("factor_variant", variant)
        g.prepare()
        g.run(2, tot, W)
        t = time.time(); g.run(W + 1, tot, tot); dt = time.time() - t
        print("%d chain(s) factor_variant %d: %.1f us per sweep, %.0f it/s %s" % (nb, variant, 1e6 * dt / K, nb * K / dt, {k: v for k, v in ch.counters().items() if v and k != 'where'}), flush=True)
        if nb > 1: g.close()
        for c in members: c.close()
