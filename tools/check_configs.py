"""All BASELINE.json synthetic configs: a few sweeps on the GPU, first rows against the oracle (where it is cheap), throughput."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
from oracle import bnr_oracle as bo
for name, (n, V, R), steps, orows in [("cfg2", (200, 50, 5), 400, 6), ("cfg3", (500, 100, 7), 400, 4), ("cfg5", (500, 300, 10), 60, 0), ("cfg4", (2000, 200, 7), 20, 0)]:
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=20240501)
    tot = steps + 1
    ch = bnr_amd.Chain(X, y, R, tot, 20240501, 1)
    ch.init_prior()
    ch.run(2, tot, 9)
    t0 = time.time(); ch.run(10, tot, tot); dt = time.time() - t0
    g = ch.fetch(1, max(orows, 2))
    msg = ""
    if orows:
        o = bo.Oracle(X, y, R, orows, 20240501, chain=1, pdf_mode=1); o.init_prior(); o.run(2, orows, orows)
        worst = max(float(np.max(np.abs(g[k][:orows] - o.t[k]) / (1e-9 + np.abs(o.t[k])))) for k in bo.COLUMNS if k not in ("xi", "lam"))
        same = all(np.array_equal(g[k][:orows], o.t[k]) for k in ("xi", "lam"))
        msg = "first %d rows vs oracle: worst rel err %.2e, discrete equal %s" % (orows, worst, same)
    full = ch.fetch(tot, tot)
    finite = all(np.all(np.isfinite(full[k])) for k in bo.COLUMNS)
    print("%s n=%d V=%d R=%d: %.1f it/s (%.0f us/it), gram standalone %.1f us, last row finite %s, counters %s; %s" % (
        name, n, V, R, (tot - 9) / dt, 1e6 * dt / (tot - 9), ch.debug_time_gram(10), finite, ch.counters(), msg), flush=True)
    ch.close()
