import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
np.set_printoptions(linewidth=200, precision=4, suppress=True)
n, V, R = 40, 8, 3
X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=4242)
ch = bnr_amd.Chain(X, y, R, 6, 4242, 1)
ch.init_prior()
ch.update("tau2", 2, 2); ch.update("u_xi", 2, 2)
try:
    ch.update("gamma", 2, 2)
except Exception as e:
    print("ERR", e)
npad = 64; ld = 2 * npad + 32
E = ch.debug_copy(0, ld * npad).reshape(npad, ld).T      # E[row, col]
b = ch.debug_copy(1, npad)
t = ch.fetch(1, 2)
S = t["S"][0, :, 0]
G = (X * S) @ X.T + np.eye(n)
Gp = np.eye(npad); Gp[:n, :n] = G
L = np.linalg.cholesky(Gp)
Lg = np.tril(E[:npad, :npad])
print("L diff by 32-block (rows, cols):")
for i in range(2):
    for j in range(i + 1):
        print(i, j, np.abs(Lg[32*i:32*i+32, 32*j:32*j+32] - L[32*i:32*i+32, 32*j:32*j+32]).max())
Y = E[npad:2*npad, :npad]
Yref = np.linalg.inv(L).T
for i in range(2):
    for j in range(i, 2):
        print("Y", i, j, np.abs(Y[32*i:32*i+32, 32*j:32*j+32] - Yref[32*i:32*i+32, 32*j:32*j+32]).max())
w = E[2*npad, :npad]
print("w diff", np.abs(w - np.linalg.solve(L, b)).max())
print("L00 gpu first rows\n", Lg[:4, :4], "\nref\n", L[:4, :4])
d = np.abs(Lg[:32, :32] - L[:32, :32])
print("L00 err per column", np.round(d.max(axis=0), 3))
print("L00 err per row", np.round(d.max(axis=1), 3))
d = np.abs(Lg[32:64, :32] - L[32:64, :32])
print("L10 err per column", np.round(d.max(axis=0), 3))
