"""Diagnostic (-DBNR_STAMPS build): phases of k_backproj block 7 of chain 1 (shader cycles): dot products | GIG draws | partial sums."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
for nb in (1, 8):
    chains = [bnr_amd.Chain(X, y, 7, 40, 20240501, 1)]
    chains += [bnr_amd.Chain.like(chains[0], 20240501, c, 40) for c in range(2, nb + 1)]
    for ch in chains: ch.init_prior()
    r = bnr_amd.Group(chains) if nb > 1 else chains[0]
    r.run(2, 40, 40)
    d = chains[0].debug_read(330).astype(np.int64)[320:324]
    print("chains %d: dots %d | GIG %d | sums %d | total %d cycles (%.2f us)" % (nb, d[1]-d[0], d[2]-d[1], d[3]-d[2], d[3]-d[0], (d[3]-d[0]) / 2400.0))
    if nb > 1: r.close()
    for ch in chains: ch.close()
