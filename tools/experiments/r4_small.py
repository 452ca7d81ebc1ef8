#!/usr/bin/env python3
"""Round 4: the one-launch factorization of small problems (k_chol_small, factor_variant 5 / the choice by size for n_pad <= 128) against the
launch-per-panel one (0): bitwise equal tables over shapes and group sizes, time per sweep."""
import sys, os, time
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
for (n, V, R) in [(70, 19, 5), (128, 12, 3), (64, 9, 2), (100, 30, 7), (1, 5, 2), (33, 2, 1), (65, 8, 4), (127, 20, 6)]:
    for nb in (1, 3, 8):
        t0, c0 = tables(n, V, R, 0, nb)
        t1, c1 = tables(n, V, R, 5, nb)
        for i, (a, b) in enumerate(zip(t0, t1)):
            for k in bo.COLUMNS:
                if not np.array_equal(a[k], b[k], equal_nan=True):
                    print("MISMATCH", (n, V, R), "chains", nb, "member", i, k, float(np.nanmax(np.abs(a[k] - b[k])))); bad += 1
        print((n, V, R), "chains", nb, "ok" if not bad else "", {k: v for k, v in c1.items() if v and k != 'where'}, flush=True)
print("mismatches:", bad)
K, W = 2000, 200
for (n, V, R) in [(70, 19, 5), (100, 30, 7), (128, 40, 7), (64, 9, 2)]:
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=20240501)
    for nb in (1, 2, 8):
        for variant in (0, 5):
            tot = K + W
            ch = bnr_amd.Chain(X, y, R, tot, 5, 1)
            members = [ch] + [bnr_amd.Chain.like(ch, 5, c, tot) for c in range(2, nb + 1)]
            for c in members: c.init_prior()
            g = bnr_amd.Group(members) if nb > 1 else ch
            g.set_option("factor_variant", variant)
            g.prepare()
            g.run(2, tot, W)
            t = time.time(); g.run(W + 1, tot, tot); dt = time.time() - t
            print("n=%d V=%d R=%d, %d chain(s) factor_variant %d: %.1f us per sweep, %.0f it/s" % (n, V, R, nb, variant, 1e6 * dt / K, nb * K / dt), flush=True)
            if nb > 1: g.close()
            for c in members: c.close()
