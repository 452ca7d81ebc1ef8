#!/usr/bin/env python3
"""Sweep time of an 8-chain group at the headline size under different schedules (wall clock over a long run)."""
import sys, os, time
os.environ.setdefault("BNR_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesiannetworkregression.jl_amd", "csrc", "_var", "exp.so"))   # the experiments build (tools/r4_build_variants.sh)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
tot = 1300
configs = [("graph, right-looking (round 2)", {"factor_variant": 0, "pipeline": 0, "graph": 1}),
           ("eager, right-looking", {"factor_variant": 0, "pipeline": 0, "graph": 0}),
           ("eager, left-looking after the Gram", {"factor_variant": 1, "pipeline": 0, "graph": 0}),
           ("graph, left-looking BESIDE the persistent Gram", {"factor_variant": 1, "pipeline": 1, "graph": 1})]
ref = None
for name, opts in configs:
    ch = bnr_amd.Chain(X, y, 7, tot, 5, 1)
    members = [ch] + [bnr_amd.Chain.like(ch, 5, c, tot) for c in range(2, 9)]
    for c in members: c.init_prior()
    g = bnr_amd.Group(members)
    for k, v in opts.items(): g.set_option(k, v)
    g.prepare()
    g.run(2, tot, 300)
    t = time.time()
    g.run(301, tot, tot)
    dt = time.time() - t
    tab = members[3].fetch(tot, tot)
    same = ref is None or all(np.array_equal(tab[k], ref[k], equal_nan=True) for k in tab)
    if ref is None: ref = tab
    print("%-62s %.1f us per sweep, %.0f it/s  last row bitwise equal to the first config: %s  %s" % (name, 1e6 * dt / 1000, 8 * 1000 / dt, same, {k: v for k, v in ch.counters().items() if v and k != 'where'}), flush=True)
    g.close()
    for c in members: c.close()
