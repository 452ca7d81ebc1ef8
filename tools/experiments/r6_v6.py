"""factor_variant 6 (k_chol_step2p: two panels per launch, eight-wave pipeline) against the one-panel pipeline: tables to rounding, alone and in a group; then timings."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, bnr_amd
worst = 0.0
for (n, V, R) in [(70, 19, 5), (193, 30, 5), (64, 9, 2), (500, 40, 4), (500, 100, 7), (130, 12, 3)]:
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=7)
    tabs = {}
    for v in (0, 6):
        ch = bnr_amd.Chain(X, y, R, 6, 3, 1)
        mates = [bnr_amd.Chain.like(ch, 3, c, 6) for c in (2, 3)]
        solo = bnr_amd.Chain.like(ch, 3, 1, 6)
        for c in [ch, solo] + mates: c.init_prior()
        g = bnr_amd.Group([mates[0], ch, mates[1]])
        g.set_option("factor_variant", v); solo.set_option("factor_variant", v)
        g.run(2, 6, 6); solo.run(2, 6, 6)
        tabs[v] = (ch.fetch(), solo.fetch())
        assert ch.counters()["chol_fail"] == 0 and solo.counters()["chol_fail"] == 0, (n, V, R, v, ch.counters(), solo.counters())
        g.close()
        for c in [ch, solo] + mates: c.close()
    for k in tabs[0][0]:
        assert np.array_equal(tabs[6][0][k], tabs[6][1][k]), ("group vs alone", n, V, R, k)
        a, b = tabs[6][1][k], tabs[0][1][k]
        d = float(np.max(np.abs(a - b) / (1e-300 + np.maximum(np.abs(a), np.abs(b))))) if a.size else 0.0
        worst = max(worst, d)
        assert np.allclose(a, b, rtol=1e-7, atol=1e-9), (n, V, R, k, d)
    print("n=%d V=%d R=%d: variant 6 = variant 0 to rounding (worst relative difference so far %.2e), group member bitwise the chain alone" % (n, V, R, worst), flush=True)
