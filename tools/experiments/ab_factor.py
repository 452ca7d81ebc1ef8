#!/usr/bin/env python3
"""Development check: the left-looking factorization (factor_variant 1) gives bitwise the tables of the right-looking one
(factor_variant 0), alone and in a lockstep group, over ragged shapes; plus the time per sweep of both at the headline size."""
import sys, os, time
os.environ.setdefault("BNR_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesiannetworkregression.jl_amd", "csrc", "_var", "exp.so"))   # the experiments build (tools/r4_build_variants.sh)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bnr_amd
from oracle import bnr_oracle as bo

def tables(n, V, R, variant, rows=6, extra=None):
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=7)
    ch = bnr_amd.Chain(X, y, R, rows, 3, 1)
    mates = [bnr_amd.Chain.like(ch, 3, c, rows) for c in (2, 3)]
    for c in [ch] + mates:
        c.init_prior()
    g = bnr_amd.Group([mates[0], ch, mates[1]])
    g.set_option("factor_variant", variant)
    for k, v in (extra or {}).items():
        g.set_option(k, v)
    g.run(2, rows, rows)
    solo = bnr_amd.Chain.like(ch, 3, 1, rows)
    solo.set_option("factor_variant", variant)
    for k, v in (extra or {}).items():
        solo.set_option(k, v)
    solo.init_prior()
    solo.run(2, rows, rows)
    out = (ch.fetch(), solo.fetch(), ch.counters(), solo.counters())
    g.close()
    for c in [ch, solo] + mates:
        c.close()
    return out

bad = 0
for (n, V, R) in [(70, 19, 5), (500, 100, 7), (130, 12, 3), (64, 9, 2), (193, 30, 5), (1, 5, 2), (33, 2, 1), (1000, 40, 4), (2000, 12, 3)]:
    t0 = tables(n, V, R, 0)
    for name, fv, extra in (("two panels per launch", 2, {}), ("left", 1, {"pipeline": 0}), ("left+persistent gram", 1, {"pipeline": 0, "gram_variant": 9}), ("pipelined", 1, {"pipeline": 1}),
                            ("static resident Gram", 0, {"gram_variant": 11}), ("static resident Gram, queue order", 0, {"gram_variant": 14}),
                            ("resident Gram, per-CU lists, cu 7 reserved", 0, {"gram_variant": 13, "resv_mask": 0x80}), ("critical chain on the origin queue", 0, {"crit_origin": 1}),
                            ("group back-projection", 0, {"group_backproj": 1})):
        t1 = tables(n, V, R, fv, extra=extra)
        for k in bo.COLUMNS:
            for i, what in ((0, "group"), (1, "alone")):
                if not np.array_equal(t0[i][k], t1[i][k], equal_nan=True):
                    d = np.nanmax(np.abs(t0[i][k] - t1[i][k]))
                    print("MISMATCH", name, (n, V, R), what, k, d); bad += 1
            if not np.array_equal(t1[0][k], t1[1][k], equal_nan=True):
                print("MISMATCH group vs alone", name, (n, V, R), k); bad += 1
    print((n, V, R), "counters", {k: v for k, v in t1[2].items() if v}, {k: v for k, v in t1[3].items() if v}, flush=True)
print("mismatches:", bad)

# timing at the headline size, 8 chains
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
for variant, pipe in ((0, 0), (2, 0), (1, 0)):
    tot = 1500
    ch = bnr_amd.Chain(X, y, 7, tot, 5, 1)
    members = [ch] + [bnr_amd.Chain.like(ch, 5, c, tot) for c in range(2, 9)]
    for c in members:
        c.init_prior()
    g = bnr_amd.Group(members)
    g.set_option("factor_variant", variant)
    g.set_option("pipeline", pipe)
    g.prepare()
    g.run(2, tot, 300)
    t = time.time()
    g.run(301, tot, tot)
    dt = time.time() - t
    print("factor_variant", variant, "pipeline", pipe, "8 chains: %.1f us per sweep, %.0f it/s" % (1e6 * dt / 1200, 8 * 1200 / dt), ch.counters(), flush=True)
    g.close()
    for c in members:
        c.close()
sys.exit(1 if bad else 0)
