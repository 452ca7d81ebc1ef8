#!/usr/bin/env python3
"""Sweep time under option sets, for 1 and 8 chains: tools/variant_time.py key=value[,key=value] ...
(headline size; BNR_SHAPE=n,V,R BNR_SWEEPS=k BNR_GROUPS=1,8 in the environment select another)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
n_, V_, R_ = (int(v) for v in os.environ.get("BNR_SHAPE", "500,100,7").split(","))
K_ = int(os.environ.get("BNR_SWEEPS", "2000"))
X, y, _ = bnr_amd.make_synthetic(n_, V_, R_, seed=20240501)
W_ = max(20, K_ // 7)
tot = K_ + W_
for nb in (int(v) for v in os.environ.get("BNR_GROUPS", "1,8").split(",")):
    for spec in sys.argv[1:]:
        opts = dict(kv.split("=") for kv in spec.split(",") if kv)
        ch = bnr_amd.Chain(X, y, R_, tot, 5, 1)
        members = [ch] + [bnr_amd.Chain.like(ch, 5, c, tot) for c in range(2, nb + 1)]
        for c in members: c.init_prior()
        g = bnr_amd.Group(members) if nb > 1 else ch
        for k, v in opts.items(): g.set_option(k, int(v))
        g.prepare()
        g.run(2, tot, W_)
        t = time.time()
        g.run(W_ + 1, tot, tot)
        dt = time.time() - t
        print("%d chain(s) %-40s %.1f us per sweep, %.0f it/s" % (nb, spec, 1e6 * dt / K_, nb * K_ / dt), flush=True)
        if nb > 1: g.close()
        for c in members: c.close()
