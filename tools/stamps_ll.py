"""Diagnostic: in-kernel stamps of k_chol_ll (role A workgroup 0 of chain 1) for one chain and a lockstep group (build with -DBNR_STAMPS)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
for nb in (1, 8):
    chains = [bnr_amd.Chain(X, y, 7, 40, 20240501, 1)]
    chains += [bnr_amd.Chain.like(chains[0], 20240501, c, 40) for c in range(2, nb + 1)]
    for ch in chains: ch.init_prior()
    r = bnr_amd.Group(chains) if nb > 1 else chains[0]
    r.set_option("graph", 0)
    r.run(2, 40, 40)
    d = chains[0].debug_read(16 * 8).reshape(16, 8).astype(np.int64)
    print("chains in the launch:", nb, " shader clock (100 MHz s_memtime? units as read) per phase of k_chol_ll (role A workgroup 0 of chain 1)")
    for p in range(16):
        t = d[p]
        print("  p=%2d load+update %5d barrier %5d sweep1 %5d mid %5d sweep2 %5d store %5d total %5d" % (p, t[1]-t[0], t[2]-t[1], t[3]-t[2], t[4]-t[3], t[5]-t[4], t[6]-t[5], t[6]-t[0]))
    if nb > 1: r.close()
    for ch in chains: ch.close()
