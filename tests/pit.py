"""Probability-integral-transform (PIT) calibration of a Gibbs trace against the reference's
full conditionals.

For every transition row i-1 -> i the conditional's parameters are computed DETERMINISTICALLY
(by the CPU oracle's *_params functions, or in numpy for gamma) from the recorded rows and the
recorded draw is transformed to U(0,1) / N(0,1).  Pooled transforms are tested with
Kolmogorov-Smirnov.  Applied to the golden traces extracted from the reference's
test/data/gen_test_results.jld2 this pins the oracle's formulas, index maps and ordering
conventions to the reference's actual behaviour; applied to a trace produced by the oracle or
by the HIP sampler it checks that sampler.
"""
import numpy as np
from scipy import stats

from oracle import bnr_oracle as bo


def _as_table(tbl, tot):
    t = {}
    for k in bo.COLUMNS:
        a = np.asfortranarray(np.array(tbl[k], dtype=np.float64))
        assert a.shape[0] == tot, (k, a.shape)
        t[k] = a
    return t


def pit_trace(tbl, X, y, R, rows=None, hyper=None, use_x=True, pdf_mode=1):
    """Return dict name -> (statistic kind, p-value or z, n).  rows: iterable of 0-based rows i (transition i-1->i)."""
    hyper = dict(eta=1.01, zeta=1.0, iota=1.0, aDelta=1.0, bDelta=1.0, nu=10) | (hyper or {})
    tot = np.asarray(tbl["tau2"]).shape[0]
    t = _as_table(tbl, tot)
    n, q = X.shape
    o = bo.Oracle(X, y, R, tot, seed=0, table=t, pdf_mode=pdf_mode, **hyper)
    V = o.V
    rows = list(range(1, tot)) if rows is None else list(rows)
    P = {k: [] for k in ["tau2", "theta", "Delta", "mu", "M", "pi0", "S"]}
    Z = {k: [] for k in ["u", "gamma"]}
    xi_num = xi_den = 0.0
    lam_obs = np.zeros(3)
    lam_exp = np.zeros(3)
    Xf = np.asarray(X, dtype=np.float64)
    XtX = Xf.T @ Xf if use_x else None
    for i in rows:
        tau2 = t["tau2"][i, 0, 0]
        # ---- tau2 (needs X: residual)
        if use_x:
            a, sc = o.tau2_params(i)
            P["tau2"].append(stats.invgamma.cdf(tau2, a=a, scale=sc))
        # ---- xi, u
        for k in range(V):
            rc, w, mu_t, Lc, _ = o.node_params(i, k)
            assert rc == 0
            p1 = 1.0 - w
            xi = t["xi"][i, k, 0]
            if 0.0 < p1 < 1.0:
                xi_num += xi - p1
                xi_den += p1 * (1 - p1)
            if xi == 1.0:
                Z["u"].extend(Lc.T @ (t["u"][i, :, k] - mu_t))
            else:
                assert np.all(t["u"][i, :, k] == 0.0)
        # ---- gamma
        W = o.compute_W(i, i - 1)
        if use_x:
            Sp = t["S"][i - 1, :, 0]
            Pm = XtX / tau2 + np.diag(1.0 / (tau2 * Sp))
            rhs = Xf.T @ (y - Xf @ W - t["mu"][i - 1, 0, 0]) / tau2
            L = np.linalg.cholesky(Pm)
            m = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
            Z["gamma"].extend(L.T @ (t["gamma"][i, :, 0] - W - m))
        # ---- S | gamma_i, W, tau2_i, theta_{i-1}:  1/S ~ InvGauss(mean sqrt(psi/chi), shape psi)
        psi = t["theta"][i - 1, 0, 0]
        chi = (t["gamma"][i, :, 0] - W) ** 2 / tau2
        P["S"].extend(stats.invgauss.cdf(1.0 / t["S"][i, :, 0], mu=np.sqrt(psi / chi) / psi, scale=psi))
        # ---- theta
        a, sc = o.theta_params(i)
        P["theta"].append(stats.gamma.cdf(t["theta"][i, 0, 0], a=a, scale=sc))
        # ---- Delta
        sx = t["xi"][i, :, 0].sum()
        P["Delta"].append(stats.beta.cdf(t["Delta"][i, 0, 0], hyper["aDelta"] + sx, hyper["bDelta"] + V - sx))
        # ---- M
        Psi, df = o.M_params(i)
        P["M"].append(stats.chi2.cdf(np.trace(Psi @ np.linalg.inv(t["M"][i])), df * R))
        # ---- mu
        if use_x:
            m_, s_ = o.mu_params(i)
            P["mu"].append(stats.norm.cdf(t["mu"][i, 0, 0], m_, s_))
        # ---- lambda
        pr = o.Lambda_params(i)
        pr = pr / pr.sum(1, keepdims=True)
        lam_exp += pr.sum(0)
        for r in range(R):
            lam_obs[{0.0: 0, 1.0: 1, -1.0: 2}[t["lam"][i, r, 0]]] += 1
        # ---- pi
        for r in range(R):
            al = o.pi_alpha(i, r)
            P["pi0"].append(stats.beta.cdf(t["pi"][i, r, 0], al[0], al[1] + al[2]))
    out = {}
    for k, v in P.items():
        if len(v):
            out[k] = ("ks_uniform_p", stats.kstest(np.asarray(v), "uniform").pvalue, len(v))
    for k, v in Z.items():
        if len(v):
            out[k] = ("ks_normal_p", stats.kstest(np.asarray(v), "norm").pvalue, len(v))
    out["xi"] = ("calibration_z", xi_num / np.sqrt(max(xi_den, 1e-300)), int(len(rows) * V))
    out["lam"] = ("chi2_p", stats.chisquare(lam_obs, lam_exp * lam_obs.sum() / lam_exp.sum()).pvalue, int(lam_obs.sum()))
    out["_lam_counts"] = (lam_obs.tolist(), lam_exp.tolist())
    return out
