"""Shapes near the LDS budgets: large R*V (u staged in LDS by k_tail) and large n (n-vectors staged by k_backproj / k_solve_a4)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
for (n, V, R) in [(30, 400, 32), (40, 1500, 10), (9000, 6, 3)]:
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=1)
    ch = bnr_amd.Chain(X, y, R, 5, 1, 1, nu=R + 2)          # the inverse Wishart needs nu >= R
    ch.init_prior(); ch.run(2, 5, 5)
    t = ch.fetch()
    print("n=%d V=%d R=%d (R*V=%d): finite %s, counters %s" % (n, V, R, R * V, all(np.all(np.isfinite(t[k])) for k in t), ch.counters()))
    ch.close()
for (n, V, R) in [(10, 600, 32), (15000, 4, 2)]:
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=1)
    try:
        bnr_amd.Chain(X, y, R, 5, 1, 1, nu=R + 2); print("n=%d V=%d R=%d: accepted?!" % (n, V, R))
    except bnr_amd.BnrError as e:
        print("n=%d V=%d R=%d rejected: %s" % (n, V, R, e))
