"""BASELINE.json configs[0]: the reference's example data (examples/matrix_networks.csv + responses.csv, V=30, n=100, R=5)
through the drop-in API, against the truth the data was simulated from (examples/true_b.csv, true_xi.csv)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "examples_xy.npz"))
X, y, tb, txi = d["X"], d["y"], d["true_b"], d["true_xi"]
nburn, nsamp, C = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
t0 = time.time()
res = bnr_amd.Fit(X, y, 5, nburn=nburn, nsamples=nsamp, num_chains=C, seed=1234, x_transform=False, suppress_timer=True,
                  filename=None, psrf_cutoff=1.05, return_state=False, summary_interval=95)
dt = time.time() - t0
s = bnr_amd.Summary(res)
V = 30
est = res.summary_device["estimate"]
off = np.array([i for i, (a, b) in enumerate(zip(s.edge_coef["node1"], s.edge_coef["node2"])) if a != b])
print("fit of %d chains x %d iterations in %.2f s; max PSRF gamma %.3f xi %.3f" % (C, nburn + nsamp, dt, res.rhatgamma.max(), res.rhatxi.max()))
g = est[off]
print("corr(posterior mean off-diagonal gamma, true_b) = %.3f" % np.corrcoef(g, tb)[0, 1])
lo, hi = res.summary_device["lower_bound"][off], res.summary_device["upper_bound"][off]
print("coverage of the 95%% intervals: %.3f; rmse %.3f (true_b rms %.3f)" % (np.mean((tb >= lo) & (tb <= hi)), np.sqrt(np.mean((g - tb) ** 2)), np.sqrt(np.mean(tb ** 2))))
p = res.summary_device["probability"]
print("P(xi=1): true nodes min %.2f mean %.2f | null nodes max %.2f mean %.2f" % (p[txi == 1].min(), p[txi == 1].mean(), p[txi == 0].max(), p[txi == 0].mean()))
