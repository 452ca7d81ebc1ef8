"""Synthetic n x V adjacency data of SURVEY.md section 8(d) (the workload of BASELINE.json's configs 2-5).

Truth follows the model's own generative structure: R* = min(R,5) latent dims, node inclusion xi*_v ~ Bernoulli(0.3),
u*_v ~ N(0, I) xi*_v, lambda*_r = (-1)^(r+1), B* = lowtri(u*' L* u*); X rows = column-wise lower triangle (incl.
diagonal) of a symmetric matrix with entries Bernoulli(0.5)*|N(0,1)| (as examples/matrix_networks.csv: ~50 % zeros);
y = mu* + X B* + N(0, tau*^2), mu* = 10, tau* = 1."""
import numpy as np


def make_synthetic(n, V, R, seed=20240501, normal_x=False):
    rng = np.random.default_rng(seed)
    q = V * (V + 1) // 2
    Rs = min(R, 5)
    xi = rng.random(V) < 0.3
    u = rng.standard_normal((Rs, V)) * xi[None, :]
    lam = np.array([(-1.0) ** r for r in range(Rs)])          # r = 1.. -> +1, -1, ...
    Bfull = u.T @ (lam[:, None] * u)
    idx_l, idx_k = [], []
    for k in range(V):
        for l in range(k, V):
            idx_l.append(l)
            idx_k.append(k)
    idx_l, idx_k = np.array(idx_l), np.array(idx_k)
    B = Bfull[idx_l, idx_k]
    if normal_x:
        X = rng.standard_normal((n, q))
    else:
        X = (rng.random((n, q)) < 0.5) * np.abs(rng.standard_normal((n, q)))
    y = 10.0 + X @ B + rng.standard_normal(n)
    return np.asfortranarray(X), y, dict(B=B, xi=xi.astype(float), V=V, q=q)
