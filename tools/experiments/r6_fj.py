"""flag_join = 1 (the sweep's join as a flag polled by k_solve_w, the critical chain on one queue) against the graph-edge join: bitwise the same tables."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, bnr_amd
for (n, V, R, C, tot) in [(193, 30, 5, 1, 40), (500, 40, 4, 3, 30), (500, 100, 7, 8, 40), (500, 100, 7, 1, 60), (70, 19, 5, 2, 50), (200, 50, 5, 1, 80)]:
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=7)
    tabs = {}
    for v in (0, 1):
        for graph in (1, 0):
            chains = [bnr_amd.Chain(X, y, R, tot, 3, 1)]
            chains += [bnr_amd.Chain.like(chains[0], 3, c, tot) for c in range(2, C + 1)]
            for c in chains: c.init_prior()
            r = bnr_amd.Group(chains) if C > 1 else chains[0]
            r.set_option("flag_join", v); r.set_option("graph", graph)
            r.run(2, tot // 2, tot // 2); r.run(tot // 2 + 1, tot, tot)
            tabs[(v, graph)] = [c.fetch() for c in chains]
            cnt = [c.counters() for c in chains]
            assert all(k["chol_fail"] == 0 for k in cnt), (n, V, R, C, v, cnt)
            if C > 1: r.close()
            for c in chains: c.close()
    for key in tabs:
        for a, b in zip(tabs[(0, 1)], tabs[key]):
            for k in a:
                assert np.array_equal(a[k], b[k]), (n, V, R, C, key, k, float(np.max(np.abs(a[k] - b[k]))))
    print("n=%d V=%d R=%d chains=%d rows=%d: flag join = graph-edge join, graph replay and eager, bitwise" % (n, V, R, C, tot), flush=True)
