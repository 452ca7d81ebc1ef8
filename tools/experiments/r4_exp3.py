#!/usr/bin/env python3
"""Round 4: what would a faster factorization buy?  The panel steps p >= N return at once (timing only, -DBNR_EXPERIMENTS build): sweep time of the
8-chain group and of one chain against N, with and without the scalar branch."""
import sys, os, time
os.environ.setdefault("BNR_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesiannetworkregression.jl_amd", "csrc", "_var", "exp.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
K, W = 800, 200
n_, V_, R_ = (int(v) for v in os.environ.get("BNR_SHAPE", "500,100,7").split(","))      # BNR_SHAPE=n,V,R selects another size
X, y, _ = bnr_amd.make_synthetic(n_, V_, R_, seed=20240501)
print("shape n=%d V=%d R=%d" % (n_, V_, R_))
L = bnr_amd.lib()
for nb in (8, 1):
    tot = W + K * 12 + 1
    ch = bnr_amd.Chain(X, y, R_, tot, 5, 1)
    members = [ch] + [bnr_amd.Chain.like(ch, 5, c, tot) for c in range(2, nb + 1)]
    for c in members: c.init_prior()
    g = bnr_amd.Group(members) if nb > 1 else ch
    g.prepare()
    g.run(2, tot, W)
    row = W
    for flags, label in ((0, "as it is"), (1, "scalar branch returns at once"), (12 << 8, "panel steps >= 12 return at once"), (8 << 8, "panel steps >= 8 return at once"),
                         (4 << 8, "panel steps >= 4 return at once"), (1 << 8, "panel steps >= 1 return at once"), ((8 << 8) | 1, "steps >= 8 and the scalar branch return at once"),
                         ((1 << 8) | 1, "steps >= 1 and the scalar branch return at once")):
        assert L.bnr_debug_set_exp(0, flags) == 0
        t = time.time()
        try:
            g.run(row + 1, tot, row + K)
        except Exception as e:
            pass                                   # (wrong numbers on purpose: the status may say so)
        dt = time.time() - t
        row += K
        print("%d chain(s) %-52s %7.1f us per sweep" % (nb, label, 1e6 * dt / K), flush=True)
    L.bnr_debug_set_exp(0, 0)
    try:
        if nb > 1: g.close()
        for c in members: c.close()
    except Exception: pass
