#!/usr/bin/env python3
"""Development: one configuration of the sweep schedule on a small group, printed step by step (run under `timeout`).
usage: stage_test.py <factor_variant> <pipeline> <gram_variant> <graph> [n V R]"""
import sys, os
os.environ.setdefault("BNR_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesiannetworkregression.jl_amd", "csrc", "_var", "exp.so"))   # the experiments build (tools/r4_build_variants.sh)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bnr_amd
fv, pipe, gv, graph = (int(a) for a in sys.argv[1:5])
n, V, R = (int(a) for a in sys.argv[5:8]) if len(sys.argv) > 7 else (70, 19, 5)
X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=7)
print("create", flush=True)
ch = bnr_amd.Chain(X, y, R, 6, 3, 1)
mates = [bnr_amd.Chain.like(ch, 3, c, 6) for c in (2, 3)]
for c in [ch] + mates:
    c.init_prior()
print("init done", flush=True)
g = bnr_amd.Group([mates[0], ch, mates[1]])
for k, v in (("factor_variant", fv), ("pipeline", pipe), ("gram_variant", gv), ("graph", graph)):
    g.set_option(k, v)
print("run", flush=True)
g.run(2, 6, 6)
print("ran", ch.counters(), flush=True)
t = ch.fetch()
print("gamma row 6:", t["γ"][5, :3, 0] if "γ" in t else list(t)[:3], flush=True)
