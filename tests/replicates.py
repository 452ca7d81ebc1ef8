"""Replicate-distribution check of a sampler against the reference's golden end-to-end run (test infrastructure).

The reference's test/test1-generate-samples-test.jl:10-46 fits test/data/test1.csv (n=70, V=19, q=190) with R=5, nburn=200,
nsamp=200, one chain, and compares with the stored run `res2` at rtol 1e-5 -- a seed-pinned sample PATH that only Julia's RNG
can reproduce.  What can be checked without Julia is that `res2` is a typical draw of OUR sampler's distribution over such
runs: N independent chains of the same setup give the replicate distribution of every window statistic (rows 201-400), and
the golden value must lie inside its central 99 % (rank-based two-sided p >= 0.01).  After 400 iterations the chains are far
from stationarity, so these are statistics of the TRANSIENT -- they pin initialisation, every conditional and their order.
"""
import numpy as np

STAT_NAMES = (["mean_tau2", "mean_theta", "mean_mu", "mean_Delta"] + ["mean_gamma%d" % (i + 1) for i in range(5)]
              + ["mean_P_xi", "min_P_xi", "max_P_xi", "max_rhat_gamma", "max_rhat_xi", "median_log_S", "frac_lambda_nonzero",
                 "sd_gamma1", "q05_gamma1", "q95_gamma1"])


def _split_rhat_single(x):
    """convergence.jl:4-65 for ONE chain: x (nsamp, nparams) -> (nparams,)"""
    n = x.shape[0] // 2
    halves = np.stack([x[:n], x[x.shape[0] - n:]])                  # (2, n, p)
    means = halves.mean(axis=1)
    var = halves.var(axis=1, ddof=1)
    W = var.mean(axis=0)
    B = means.var(axis=0, ddof=1)
    varp = (n - 1) / n * W + B
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.sqrt(varp / W)
    r[(varp == 0) & (W == 0)] = 1.0
    return r


def window_stats(t, nburn, nsamp):
    """The window statistics of one run (a state table in the reference layout) as a vector ordered like STAT_NAMES."""
    w = slice(nburn, nburn + nsamp)
    g = t["gamma"][w, :, 0]
    xi = t["xi"][w, :, 0]
    pxi = xi.mean(axis=0)
    rx = _split_rhat_single(xi)
    rx = rx[np.isfinite(rx)]
    out = [t["tau2"][w].mean(), t["theta"][w].mean(), t["mu"][w].mean(), t["Delta"][w].mean()]
    out += list(g[:, :5].mean(axis=0))
    out += [pxi.mean(), pxi.min(), pxi.max(), np.nanmax(_split_rhat_single(g)), rx.max() if rx.size else 1.0,
            np.median(np.log(t["S"][w, :, 0])), (t["lam"][w, :, 0] != 0).mean(), g[:, 0].std(ddof=1),
            np.sort(g[:, 0])[int(round(0.05 * nsamp)) - 1], np.sort(g[:, 0])[int(round(0.95 * nsamp)) - 1]]
    return np.array(out, dtype=np.float64)


def rank_pvalues(golden, reps):
    """Two-sided rank p-value of every golden statistic within its N replicates: 2 min(r + 1, N - r + 1) / (N + 1), r = number
    of replicates below the golden value (the golden run counts as one more draw of the same distribution)."""
    reps = np.asarray(reps)
    N = reps.shape[0]
    r = (reps < golden[None, :]).sum(axis=0)
    return np.minimum(1.0, 2.0 * np.minimum(r + 1, N - r + 1) / (N + 1))


def assert_golden_is_typical(golden, reps, p_min=0.01, what=""):
    p = rank_pvalues(golden, reps)
    bad = [(STAT_NAMES[i], float(golden[i]), float(np.min(reps[:, i])), float(np.median(reps[:, i])), float(np.max(reps[:, i])), float(p[i]))
           for i in range(len(STAT_NAMES)) if p[i] < p_min]
    assert not bad, "%s golden res2 outside the central %.0f %% of %d replicates: %s" % (what, 100 * (1 - p_min), reps.shape[0], bad)
    return p
