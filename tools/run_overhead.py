"""Fixed cost of a run call: wall time of 20-sweep and 200-sweep run calls of the 8-chain group against the per-sweep time of a long call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
tot = 3000
ch = bnr_amd.Chain(X, y, 7, tot, 5, 1)
members = [ch] + [bnr_amd.Chain.like(ch, 5, c, tot) for c in range(2, 9)]
for c in members: c.init_prior()
g = bnr_amd.Group(members)
for kv in sys.argv[1:]:
    k, v = kv.split("="); g.set_option(k, int(v))
g.prepare()
g.run(2, tot, 300)
pos = 300
def timed(n):
    global pos
    t = time.perf_counter(); g.run(pos + 1, tot, pos + n); dt = time.perf_counter() - t
    pos += n
    return dt
long = timed(1000) / 1000
for n in (20, 20, 20, 50, 200):
    dt = timed(n)
    print("%4d sweeps: %.1f us per sweep; fixed cost of the call %.0f us (long call: %.1f us per sweep)" % (n, 1e6 * dt / n, 1e6 * (dt - n * long), 1e6 * long))
