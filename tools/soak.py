"""Long run at the headline size through the drop-in API: 8 chains x 70 000 iterations with the purge ring (the plan of one run!
call grows past its first allocation), then the reference's default Fit! sizes (30 000 + 20 000).  Checks the failure counters,
prints Rhat / ESS and the wall time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
X, y, truth = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
for (nburn, nsamp, purge) in ((60000, 10000, 1000), (30000, 20000, None)):
    keep = []
    t0 = time.time()
    res = bnr_amd.generate_samples(X, y, 7, nburn=nburn, nsamp=nsamp, maxburn=nburn, psrf_cutoff=50.0, x_transform=False, suppress_timer=True,
                                   num_chains=8, seed=4242, purge_burn=purge, _keep=keep, return_state=False, summary_interval=95, ess_max_lag=0)
    dt = time.time() - t0
    cs = keep[0]
    cnt = [cs.chains[c].counters() for c in cs.ids]
    print("nburn %d nsamp %d purge %s: %.1f s wall (%.0f it/s incl. setup, Rhat, ESS, Summary); max Rhat gamma %.3f xi %.3f; ESS gamma min %.0f median %.0f of %d draws"
          % (nburn, nsamp, purge, dt, 8 * (nburn + nsamp) / dt, res.rhatgamma.max(), res.rhatxi.max(), np.nanmin(res.essgamma), np.nanmedian(res.essgamma), 8 * nsamp))
    print("   counters (sum over chains): jitter %d nan_w %d sampler_cap %d chol_fail %d" % tuple(sum(c[k] for c in cnt) for k in ("jitter", "nan_w", "sampler_cap", "chol_fail")))
    B = truth["B"]
    est = res.summary_device["estimate"]
    print("   corr(posterior mean gamma of chain 1, B*) = %.3f; P(xi=1) true nodes min %.2f, null nodes max %.2f" % (
        np.corrcoef(est, B)[0, 1], res.summary_device["probability"][truth["xi"] == 1].min(), res.summary_device["probability"][truth["xi"] == 0].max()))
    cs.close()
