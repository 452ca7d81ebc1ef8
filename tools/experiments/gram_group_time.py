#!/usr/bin/env python3
"""Gram launch time of an 8-chain group (HIP events around the launch, eager) for kernel variants / schedules."""
import sys, os, time
os.environ.setdefault("BNR_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesiannetworkregression.jl_amd", "csrc", "_var", "exp.so"))   # the experiments build (tools/r4_build_variants.sh)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
tot = 64
ch = bnr_amd.Chain(X, y, 7, tot, 5, 1)
members = [ch] + [bnr_amd.Chain.like(ch, 5, c, tot) for c in range(2, 9)]
for c in members: c.init_prior()
g = bnr_amd.Group(members)
for name, opts in (("k_gram8 single stream", {"gram_variant": 8, "overlap": 0, "pipeline": 0}),
                   ("k_gram8p single stream (nothing reserved)", {"gram_variant": 9, "overlap": 0, "pipeline": 0}),
                   ("k_gram8 two branches", {"gram_variant": 8, "overlap": 1, "pipeline": 0}),
                   ("pipelined (k_gram8p, reserved CUs)", {"gram_variant": 0, "overlap": 1, "pipeline": 1})):
    for k, v in opts.items(): g.set_option(k, v)
    g.set_profiling(True)
    g.run(2, tot, tot)
    us, n = g.last_timing(1)
    it, _ = g.last_timing(0)
    print("%-48s gram %.1f us (%d launches), sweep %.1f us" % (name, us, n, it), flush=True)
    g.set_profiling(False)
print(ch.counters())
