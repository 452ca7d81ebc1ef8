"""chol_multi = 1 (k_chol_steps: several panel steps per launch, barrier among the chain's workgroups on their XCD) against the launch-per-step factorization: bitwise the same
tables (same arithmetic, same order), alone and in groups of 3 and 8; then us per sweep."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, bnr_amd
for (n, V, R, C) in [(193, 30, 5, 1), (500, 40, 4, 3), (500, 100, 7, 8), (500, 100, 7, 1), (130, 12, 3, 2), (320, 20, 3, 5)]:
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=7)
    tabs = {}
    for v in (0, 1):
        chains = [bnr_amd.Chain(X, y, R, 8, 3, 1)]
        chains += [bnr_amd.Chain.like(chains[0], 3, c, 8) for c in range(2, C + 1)]
        for c in chains: c.init_prior()
        r = bnr_amd.Group(chains) if C > 1 else chains[0]
        r.set_option("chol_multi", v)
        t0 = time.perf_counter()
        r.run(2, 8, 8)
        dt = time.perf_counter() - t0
        tabs[v] = [c.fetch() for c in chains]
        cnt = [c.counters() for c in chains]
        assert all(k["chol_fail"] == 0 for k in cnt), (n, V, R, C, v, cnt)
        if C > 1: r.close()
        for c in chains: c.close()
        print("  n=%d V=%d R=%d chains=%d chol_multi=%d: 7 sweeps (first call, with graph capture) in %.3f s" % (n, V, R, C, v, dt), flush=True)
    for a, b in zip(tabs[0], tabs[1]):
        for k in a:
            assert np.array_equal(a[k], b[k]), (n, V, R, C, k, float(np.max(np.abs(a[k] - b[k]))))
    print("n=%d V=%d R=%d chains=%d: bitwise equal" % (n, V, R, C), flush=True)
