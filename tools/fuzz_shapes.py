"""Randomised parity sweep: random (n, V, R, seed, X flavour, hyper-parameters), a few Gibbs sweeps on the GPU (alone or as a
member of a lockstep group) against the CPU oracle on identical variates.  usage: fuzz_shapes.py <cases> [seed] [scale | binary]   (BNR_FUZZ_OPTS=name=value,...: chain / group options set before the run)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
from oracle import bnr_oracle as bo
N = int(sys.argv[1]); rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
SCALE = len(sys.argv) > 3 and sys.argv[3] == "scale"
BINARY = len(sys.argv) > 3 and sys.argv[3] == "binary"     # Bool model matrix, the Gram forced onto the i8 matrix pipe (round 5) whatever the size
worst_all = 0.0
for case in range(N):
    V = int(rng.integers(2, 41)); R = int(rng.integers(1, 13)); n = int(rng.choice([1, 2, 3, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 200, int(rng.integers(4, 260)), int(rng.integers(260, 700))]))
    tot = int(rng.integers(3, 9)); seed = int(rng.integers(1, 10**6)); normal_x = bool(rng.integers(0, 2)); group = bool(rng.integers(0, 2))
    hyper = dict(eta=float(rng.choice([1.01, 0.5, 2.0])), zeta=float(rng.choice([1.0, 0.3])), iota=float(rng.choice([1.0, 2.5])),
                 aDelta=float(rng.choice([1.0, 0.0, 3.0])), bDelta=float(rng.choice([1.0, 0.0, 2.0])), nu=float(max(R, rng.choice([10, 12, R + 1]))))
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=seed, normal_x=normal_x)
    if SCALE:                                                  # badly scaled data: X and y over several orders of magnitude
        sx, sy = 10.0 ** rng.integers(-3, 4), 10.0 ** rng.integers(-3, 4)
        X = np.asfortranarray(X * sx); y = y * sy
    if BINARY:
        Xb = np.asfortranarray(rng.random(X.shape) < float(rng.choice([0.5, 0.1, 0.9])))
        X = np.asfortranarray(Xb.astype(np.float64))           # what the oracle sees
    ch = bnr_amd.Chain(bnr_amd.XInput(Xb, False) if BINARY else X, y, R, tot, seed, 1, **hyper)
    if BINARY:
        ch.set_option("gram_i8", 1)
        assert ch.last_timing(4)[0] == 1
    gsize = int(rng.integers(2, 8))
    mates = [bnr_amd.Chain.like(ch, seed, c, tot) for c in range(2, gsize + 1)] if group else []
    if BINARY:
        for c in mates: c.set_option("gram_i8", 1)
    for c in [ch] + mates: c.init_prior()
    g = bnr_amd.Group(mates[:1] + [ch] + mates[1:]) if group else None
    for kv in os.environ.get("BNR_FUZZ_OPTS", "").split(","):        # e.g. BNR_FUZZ_OPTS=wide_backproj=1: the whole sweep under a non-default kernel choice
        if kv: (g or ch).set_option(kv.split("=")[0], int(kv.split("=")[1]))
    (g or ch).run(2, tot, tot)
    got = ch.fetch()
    o = bo.Oracle(X, y, R, tot, seed, chain=1, pdf_mode=1, **hyper); o.init_prior(); o.run(2, tot, tot)
    worst = 0.0
    yard = 0.0
    if SCALE:
        # badly scaled data makes X D X' + I ill-conditioned (cond up to 1e10): the yardstick is how far the oracle's own two
        # equivalent solve paths (the reference's LU + full GEMM vs Cholesky) drift apart on the same variates
        o2 = bo.Oracle(X, y, R, tot, seed, chain=1, pdf_mode=1, cost_mode=1, **hyper); o2.init_prior(); o2.run(2, tot, tot)
        for k in bo.COLUMNS:
            if k not in ("xi", "lam"):
                yard = max(yard, float(np.max(np.abs(o2.t[k] - o.t[k]) / (1e-9 + np.abs(o.t[k])))))
        if not all(np.array_equal(o2.t[k], o.t[k]) for k in ("xi", "lam")):
            print("case %d: the oracle's own LU and Cholesky paths flip a discrete draw (n=%d V=%d R=%d) -- skipped" % (case, n, V, R)); ch.close(); [c.close() for c in mates]; continue
    for k in bo.COLUMNS:
        a, b = got[k], o.t[k]
        if k in ("xi", "lam"):
            assert np.array_equal(a, b), (case, k, n, V, R)
        else:
            worst = max(worst, float(np.max(np.abs(a - b) / (1e-9 + np.abs(b)))))
    cnt = ch.counters()
    assert worst < max(1e-6, 30.0 * yard) and cnt["chol_fail"] == 0, (case, n, V, R, worst, yard, cnt)
    worst_all = max(worst_all, worst)
    if g: g.close()
    for c in [ch] + mates: c.close()
    if case % 20 == 19: print("case %d ok (n=%d V=%d R=%d group=%s), worst so far %.2e" % (case + 1, n, V, R, group, worst_all), flush=True)
print(("BINARY model matrices, Gram on the i8 pipe: " if BINARY else "") + "all %d cases within %s of the oracle (worst relative error %.2e); discrete columns equal" % (N, "max(1e-6, 30 x the oracle's own LU-vs-Cholesky drift)" if SCALE else "1e-6", worst_all))
