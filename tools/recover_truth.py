"""Exploration: does the sampler recover a synthetic truth?  (group of chains on the GPU)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
n, V, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nburn, nsamp, C = int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
X, y, truth = bnr_amd.make_synthetic(n, V, R, seed=int(sys.argv[7]) if len(sys.argv) > 7 else 11)
tot = nburn + nsamp
chains = [bnr_amd.Chain(X, y, R, tot, 99, 1)]
chains += [bnr_amd.Chain.like(chains[0], 99, c, tot) for c in range(2, C + 1)]
for ch in chains: ch.init_prior()
g = bnr_amd.Group(chains) if C > 1 else chains[0]
t0 = time.time(); g.run(2, nburn, tot); dt = time.time() - t0
stats = np.stack([ch.rhat_stats(nburn + 1, nsamp) for ch in chains])
rh = bnr_amd.rhat_from_stats(stats, nsamp)
q = V * (V + 1) // 2
means = np.stack([ch.summary(nburn + 1, nsamp, max(1, round(nsamp * 0.025)), round(nsamp * 0.975)) [0] for ch in chains])
pxi = np.stack([ch.summary(nburn + 1, nsamp, 1, nsamp)[3] for ch in chains])
lo = np.stack([ch.summary(nburn + 1, nsamp, max(1, round(nsamp * 0.025)), round(nsamp * 0.975))[1] for ch in chains]).mean(0)
hi = np.stack([ch.summary(nburn + 1, nsamp, max(1, round(nsamp * 0.025)), round(nsamp * 0.975))[2] for ch in chains]).mean(0)
m = means.mean(0)
B = truth["B"]
print("%d chains x %d it in %.2fs (%.0f it/s); max rhat gamma %.3f xi %.3f" % (C, tot, dt, C * tot / dt, np.nanmax(rh[:q]), np.nanmax(rh[q:])))
print("corr(mean gamma, B*) = %.3f; rmse %.3f (B* rms %.3f); coverage of 95%% intervals %.2f" % (np.corrcoef(m, B)[0, 1], np.sqrt(np.mean((m - B) ** 2)), np.sqrt(np.mean(B ** 2)), np.mean((B >= lo) & (B <= hi))))
p = pxi.mean(0)
print("P(xi=1) for true nodes: %s" % np.round(p[truth["xi"] == 1], 2))
print("P(xi=1) for null nodes: %s" % np.round(p[truth["xi"] == 0], 2))
print("counters", chains[0].counters())
