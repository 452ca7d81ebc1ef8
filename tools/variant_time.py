#!/usr/bin/env python3
"""Sweep time under option sets, for 1 and 8 chains at the headline size: tools/variant_time.py key=value[,key=value] ..."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
tot = 2300
for nb in (1, 8):
    for spec in sys.argv[1:]:
        opts = dict(kv.split("=") for kv in spec.split(",") if kv)
        ch = bnr_amd.Chain(X, y, 7, tot, 5, 1)
        members = [ch] + [bnr_amd.Chain.like(ch, 5, c, tot) for c in range(2, nb + 1)]
        for c in members: c.init_prior()
        g = bnr_amd.Group(members) if nb > 1 else ch
        for k, v in opts.items(): g.set_option(k, int(v))
        g.prepare()
        g.run(2, tot, 300)
        t = time.time()
        g.run(301, tot, tot)
        dt = time.time() - t
        print("%d chain(s) %-40s %.1f us per sweep, %.0f it/s" % (nb, spec, 1e6 * dt / 2000, nb * 2000 / dt), flush=True)
        if nb > 1: g.close()
        for c in members: c.close()
