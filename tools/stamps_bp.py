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
    full = chains[0].debug_read(340).astype(np.int64)
    d, g = full[320:324], full[330:335]
    print("chains %d, per-chain kernel: dots %d | GIG %d | sums %d | total %d cycles (%.2f us)" % (nb, d[1]-d[0], d[2]-d[1], d[3]-d[2], d[3]-d[0], (d[3]-d[0]) / 2400.0))
    if g[4] > g[0] > 0: print("chains %d, group kernel: staging %d | dots %d | GIG %d | sums %d | total %d cycles (%.2f us)" % (nb, g[1]-g[0], g[2]-g[1], g[3]-g[2], g[4]-g[3], g[4]-g[0], (g[4]-g[0]) / 2400.0))
    if nb > 1: r.close()
    for ch in chains: ch.close()
