"""Re-run one case of tools/fuzz_shapes.py (same rng stream) and show where GPU and oracle differ; also the oracle against
itself with a different (mathematically equivalent) solve path, as a yardstick for the conditioning of the case."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
from oracle import bnr_oracle as bo
target = int(sys.argv[1]); rng = np.random.default_rng(int(sys.argv[2])); SCALE = len(sys.argv) > 3
for case in range(target + 1):
    V = int(rng.integers(2, 41)); R = int(rng.integers(1, 13)); n = int(rng.choice([1, 2, 3, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 200, int(rng.integers(4, 260)), int(rng.integers(260, 700))]))
    tot = int(rng.integers(3, 9)); seed = int(rng.integers(1, 10**6)); normal_x = bool(rng.integers(0, 2)); group = bool(rng.integers(0, 2))
    hyper = dict(eta=float(rng.choice([1.01, 0.5, 2.0])), zeta=float(rng.choice([1.0, 0.3])), iota=float(rng.choice([1.0, 2.5])),
                 aDelta=float(rng.choice([1.0, 0.0, 3.0])), bDelta=float(rng.choice([1.0, 0.0, 2.0])), nu=float(max(R, rng.choice([10, 12, R + 1]))))
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=seed, normal_x=normal_x)
    sx = sy = 1.0
    if SCALE:
        sx, sy = 10.0 ** rng.integers(-3, 4), 10.0 ** rng.integers(-3, 4)
        X = np.asfortranarray(X * sx); y = y * sy
    gsize = int(rng.integers(2, 8))
print("case", target, "n", n, "V", V, "R", R, "tot", tot, "scale X", sx, "scale y", sy, hyper)
ch = bnr_amd.Chain(X, y, R, tot, seed, 1, **hyper); ch.init_prior(); ch.run(2, tot, tot); got = ch.fetch()
o = bo.Oracle(X, y, R, tot, seed, chain=1, pdf_mode=1, **hyper); o.init_prior(); o.run(2, tot, tot)
o2 = bo.Oracle(X, y, R, tot, seed, chain=1, pdf_mode=1, cost_mode=1, **hyper); o2.init_prior(); o2.run(2, tot, tot)
for k in bo.COLUMNS:
    e1 = np.abs(got[k] - o.t[k]) / (1e-9 + np.abs(o.t[k])); e2 = np.abs(o2.t[k] - o.t[k]) / (1e-9 + np.abs(o.t[k]))
    print("%-6s GPU vs oracle: worst %.2e at row %d | oracle (LU, full GEMM) vs oracle (Cholesky): worst %.2e" % (k, e1.max(), int(np.unravel_index(e1.argmax(), e1.shape)[0]) + 1, e2.max()))
G = (X * o.t["S"][0, :, 0]) @ X.T + np.eye(n)
print("cond(X D X' + I) at row 1: %.2e" % np.linalg.cond(G))
