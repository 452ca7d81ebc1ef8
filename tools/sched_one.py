import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
tot = 60
opts = dict(a.split("=") for a in sys.argv[1:])
ch = bnr_amd.Chain(X, y, 7, tot, 5, 1)
members = [ch] + [bnr_amd.Chain.like(ch, 5, c, tot) for c in range(2, 9)]
for c in members: c.init_prior()
g = bnr_amd.Group(members)
for k, v in opts.items(): g.set_option(k, int(v))
g.run(2, tot, tot)
print(ch.counters())
