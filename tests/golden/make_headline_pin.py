"""Generates tests/golden/headline_pin.npz: posterior summaries of the CPU restatement (oracle/, the reference's algorithm: gibbs.jl:191-677, gig.jl) at the HEADLINE size --
BASELINE.json configs[2]: synthetic n=500, V=100 (q=5050), R=7, the fit's 8 chains (seed 20240501, chains 1..8), 1000 burn-in + 2000 kept sweeps per chain.
Per chain: mean of xi (V) and of gamma on a fixed subset of edges, and the same over 20 batches of 100 sweeps (batch means -> Monte-Carlo standard errors).
tests/test_gpu_parity.py::test_headline_size_posterior_summaries_match_the_cpu_restatement runs the same fit on the GPU and compares within MCSE.
~7 minutes on 8 cores (0.14 s per sweep and chain).   usage: python tests/golden/make_headline_pin.py"""
import os, sys, numpy as np
from multiprocessing import Pool
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
N, V, R, SEED, NBURN, NSAMP, NBATCH = 500, 100, 7, 20240501, 1000, 2000, 20
EDGES = np.arange(0, V * (V + 1) // 2, 13)

def one(c):
    import bnr_amd
    from oracle import bnr_oracle as bo
    X, y, _ = bnr_amd.make_synthetic(N, V, R, seed=SEED)
    tot = NBURN + NSAMP + 1
    o = bo.Oracle(X, y, R, tot, SEED, chain=c, pdf_mode=1)
    o.init_prior()
    o.run(2, tot, tot)
    xi = np.asarray(o.t["xi"])[NBURN + 1:, :, 0] if np.asarray(o.t["xi"]).ndim == 3 else np.asarray(o.t["xi"])[NBURN + 1:]
    g = np.asarray(o.t["gamma"])
    g = (g[NBURN + 1:, :, 0] if g.ndim == 3 else g[NBURN + 1:])[:, EDGES]
    tau2 = np.asarray(o.t["tau2"]).reshape(tot, -1)[NBURN + 1:, 0]
    bs = NSAMP // NBATCH
    return (xi.mean(0), g.mean(0), xi.reshape(NBATCH, bs, -1).mean(1), g.reshape(NBATCH, bs, -1).mean(1), tau2.mean(), tau2.reshape(NBATCH, bs).mean(1))

if __name__ == "__main__":
    with Pool(8) as p:
        res = p.map(one, range(1, 9))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "headline_pin.npz"),
                        n=N, V=V, R=R, seed=SEED, nburn=NBURN, nsamp=NSAMP, edges=EDGES,
                        xi_mean=np.stack([r[0] for r in res]), gamma_mean=np.stack([r[1] for r in res]),
                        xi_batch=np.stack([r[2] for r in res]).astype(np.float32), gamma_batch=np.stack([r[3] for r in res]).astype(np.float32),   # (batch means only feed the standard errors: float32)
                        tau2_mean=np.array([r[4] for r in res]), tau2_batch=np.stack([r[5] for r in res]).astype(np.float32))
    print("written; P(xi = 1) over chains and nodes: min %.3f max %.3f" % (np.stack([r[0] for r in res]).min(), np.stack([r[0] for r in res]).max()))
