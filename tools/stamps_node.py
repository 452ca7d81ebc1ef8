#!/usr/bin/env python3
"""Diagnostic (-DBNR_STAMPS build): k_node's workgroups from inside -- start, end of the neighbour sums, end, XCD -- with the launch-per-panel factorization
(factor_variant 0) and with the data-flow one (4) running beside it: tools/stamps_node.py [nchains]"""
import sys, os
os.environ.setdefault("BNR_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesiannetworkregression.jl_amd", "csrc", "_var", "stamps.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
for variant in (0, 4):
    chains = [bnr_amd.Chain(X, y, 7, 40, 20240501, 1)]
    chains += [bnr_amd.Chain.like(chains[0], 20240501, c, 40) for c in range(2, nb + 1)]
    for ch in chains: ch.init_prior()
    g = bnr_amd.Group(chains) if nb > 1 else chains[0]
    g.set_option("factor_variant", variant)
    if len(sys.argv) > 2: g.set_option("graph", int(sys.argv[2]))
    g.run(2, 40, 40)
    d = chains[0].debug_read(2400).astype(np.int64)
    st = d[2000:2400].reshape(100, 4).copy()
    hw = st[:, 3] >> 8
    st[:, 3] &= 7
    t0 = st[:, 0].min()
    cu_of = lambda h: ((h >> 13) & 7, (h >> 8) & 15, (h >> 4) & 3)          # (se, cu, simd)
    print("factor_variant %d, %d chain(s): k_node of the last sweep, member 0 (us after its first workgroup started)" % (variant, nb))
    if variant == 4:
        nw = 32
        ds, de, dh = d[1000:1000 + nw], d[1100:1100 + nw], d[1200:1200 + nw]
        print("  k_chol_df, chain 0: workgroups start %.1f .. %.1f, end %.1f .. %.1f us (same clock)" % ((ds.min() - t0) / 100.0, (ds.max() - t0) / 100.0, (de.min() - t0) / 100.0, (de.max() - t0) / 100.0))
        late = sorted((int(ds[i] - t0) / 100.0, i, cu_of(int(dh[i]))[:2]) for i in range(nw))[-4:]
        print("    the last four to start: ", late)
        dfcus = {cu_of(int(dh[i]))[:2] for i in range(nw)}
        m0 = st[:, 3] == 0
        print("    k_node's XCD-0 workgroups ran on (se, cu, simd):", sorted(cu_of(int(h)) for h in hw[m0]), " -- CUs with a k_chol_df workgroup: %d" % len(dfcus))
    for xcd in range(8):
        m = st[:, 3] == xcd
        if not m.any(): continue
        s0, s1, s2 = ((st[m, i] - t0) / 100.0 for i in range(3))
        print("  XCD %d: %2d workgroups, start %6.1f .. %6.1f, sums done after %5.1f .. %5.1f us, whole workgroup %5.1f .. %5.1f us, last end %6.1f" % (
            xcd, m.sum(), s0.min(), s0.max(), (s1 - s0).min(), (s1 - s0).max(), (s2 - s0).min(), (s2 - s0).max(), s2.max()))
    if nb > 1: g.close()
    for ch in chains: ch.close()
