import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
for nb in (2, 3, 4, 6, 8, 12, 16, 24):
    tot = 60
    ch = bnr_amd.Chain(X, y, 7, tot, 5, 1)
    members = [ch] + [bnr_amd.Chain.like(ch, 5, c, tot) for c in range(2, nb + 1)]
    for c in members: c.init_prior()
    g = bnr_amd.Group(members)
    g.set_option("overlap", 0); g.set_profiling(True)
    g.run(2, tot, tot)
    us, n = g.last_timing(1)
    print("%2d chains: %4d workgroups = %.2f rounds of 768: Gram %.1f us = %.2f us per chain" % (nb, 252 * nb, 252 * nb / 768.0, us, us / nb), flush=True)
    g.close()
    for c in members: c.close()
