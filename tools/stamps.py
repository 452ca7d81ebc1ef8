import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
ch = bnr_amd.Chain(X, y, 7, 40, 20240501, 1)
ch.set_option("graph", 0)
ch.init_prior(); ch.run(2, 40, 40)
d = ch.debug_read(16 * 8).reshape(16, 8).astype(np.int64)
print("chol_step block 0 (role A, diagonal panel): shader cycles per phase [load, update, sweep, store], total")
for p in range(16):
    t = d[p]
    print(p, "update", int(t[2]-t[0]), "sweep", int(t[3]-t[2]), "store", int(t[4]-t[3]), "total", int(t[4]-t[0]))
d = ch.debug_read(300)[256:264].astype(np.int64)
print("k_tail (thread 0) cycles: load+reduce %d, sums+Psi %d, draws %d, M/Minv(wave0) %d, qpass %d, final %d, total %d" % (d[1]-d[0], d[2]-d[1], d[3]-d[2], d[4]-d[3], d[5]-d[4], d[6]-d[5], d[6]-d[0]))
d = ch.debug_read(330)[320:324].astype(np.int64)
print("k_backproj block 7 cycles: dots %d, gamma+GIG %d, sums %d, total %d" % (d[1]-d[0], d[2]-d[1], d[3]-d[2], d[3]-d[0]))
from oracle import bnr_oracle as bo
o = bo.Oracle(X, y, 7, 6, 20240501, chain=1, pdf_mode=1); o.init_prior(); o.run(2, 6, 6)
g = ch.fetch(1, 6)
for k in bo.COLUMNS:
    print(k, float(np.max(np.abs(g[k] - o.t[k]) / (1e-9 + np.abs(o.t[k])))))
print(ch.counters())
