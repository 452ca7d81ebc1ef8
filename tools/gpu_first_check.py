"""Diagnostic (not a test): run the HIP sampler next to the CPU oracle and print per-column errors."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bnr_amd
from oracle import bnr_oracle as bo


def compare(got, ref, rows=None, tag=""):
    worst = 0
    for k in bo.COLUMNS:
        a, b = got[k], ref[k]
        if rows is not None:
            a, b = a[rows], b[rows]
        err = np.abs(a - b) / (1e-9 + np.abs(b))
        worst = max(worst, np.nanmax(err))
        print("  %s %-6s max rel err %.3e  nan=%d" % (tag, k, np.nanmax(err), np.isnan(a).sum()))
    return worst


def case(n, V, R, tot, seed, normal_x=False, hooks=True):
    print("=== case n=%d V=%d R=%d tot=%d" % (n, V, R, tot))
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=seed, normal_x=normal_x)
    ch = bnr_amd.Chain(X, y, R, tot, seed, 1, device=0)
    ch.init_prior()
    o = bo.Oracle(X, y, R, tot, seed, chain=1, pdf_mode=1)
    o.init_prior()
    g = ch.fetch(1, 1)
    print(" init row:")
    compare(g, {k: v[:1] for k, v in o.t.items()})
    if hooks:
        # deconstructed sweep on row 2 (mirrors test/init-tests.jl:96-124)
        for name in ["tau2", "u_xi", "gamma", "D", "theta", "Delta", "M", "mu", "Lambda", "pi"]:
            ch.update(name, 2, 2)
            o.update(name, 1, 2)
            g = ch.fetch(2, 2)
            cols = dict(tau2=["tau2"], u_xi=["u", "xi"], gamma=["gamma"], D=["S"], theta=["theta"], Delta=["Delta"], M=["M"],
                        mu=["mu"], Lambda=["lam"], pi=["pi"])[name]
            for k in cols:
                a, b = g[k][0], o.t[k][1]
                print("  hook %-7s %-6s max rel err %.3e" % (name, k, np.nanmax(np.abs(a - b) / (1e-9 + np.abs(b)))))
    t = time.time()
    ch.run(2, tot, tot)
    tg = time.time() - t
    t = time.time()
    o.iter = 1
    o.run(2, tot, tot)
    to = time.time() - t
    g = ch.fetch()
    print(" full run: gpu %.3fs (%.1f it/s)  oracle %.3fs (%.1f it/s)" % (tg, (tot - 1) / tg, to, (tot - 1) / to))
    for r in (1, 2, 5, tot - 1):
        if r < tot:
            w = compare(g, o.t, rows=slice(r, r + 1), tag="row%d" % (r + 1))
    print(" counters", ch.counters())
    ch.close()


if __name__ == "__main__":
    print("devices:", bnr_amd.device_count())
    case(40, 8, 3, 12, 4242)
    case(70, 19, 5, 40, 1234, normal_x=True)
    case(200, 50, 5, 20, 99, hooks=False)
    case(500, 100, 7, 8, 5, hooks=False)
