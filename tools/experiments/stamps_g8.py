"""Diagnostic: duration and start of every 16th k_gram8 workgroup (build with -DBNR_STAMPS)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
chains = [bnr_amd.Chain(X, y, 7, 8, 20240501, 1)]
chains += [bnr_amd.Chain.like(chains[0], 20240501, c, 8) for c in range(2, 9)]
for ch in chains: ch.init_prior()
g = bnr_amd.Group(chains)
for k, v in (("graph", 0), ("gram_variant", 8), ("overlap", 0), ("pipeline", 0)): g.set_option(k, v)
g.run(2, 8, 8)
d = chains[0].debug_read(1000).astype(np.int64)
t0 = min(int(d[641 + 2 * i]) for i in range(126) if d[641 + 2 * i] > 0)
print("workgroup: start us, duration us")
print("  ".join("%d: %+.0f %.0f" % (16 * i, (d[641 + 2 * i] - t0) / 100.0, d[640 + 2 * i] / 100.0) for i in range(126)))
