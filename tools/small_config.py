"""The reference's only recorded timing (test/testing-tests.md:276: 17.1 it/s on an i7-8569U) is at n=100, V=30 (q=465), R=7:
the same size on the GPU, one chain and a lockstep group of 8."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(100, 30, 7, seed=20240501)
for C in (1, 8):
    tot = 4050
    chains = [bnr_amd.Chain(X, y, 7, tot, 1, 1)]
    chains += [bnr_amd.Chain.like(chains[0], 1, c, tot) for c in range(2, C + 1)]
    for ch in chains: ch.init_prior()
    r = bnr_amd.Group(chains) if C > 1 else chains[0]
    r.run(2, 49, 49)
    t0 = time.perf_counter(); r.run(50, tot, tot); dt = time.perf_counter() - t0
    print("n=100 V=30 R=7, %d chain(s): %.0f it/s over all chains (%.1f us per sweep)" % (C, C * (tot - 49) / dt, 1e6 * dt / (tot - 49)))
    if C > 1: r.close()
    for ch in chains: ch.close()
