"""Diagnostic (-DBNR_STAMPS build): phases inside bnr_panel_sweep of panel workgroup 0 (shader cycles): stage-in + barrier | first-half loads |
first 16-column sweep | publish + barrier | MFMA mid update + barrier | second-half loads | second sweep | (hand-back + barrier = rest)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
ch = bnr_amd.Chain(X, y, 7, 40, 20240501, 1)
ch.init_prior()
ch.run(2, 40, 40)
d = ch.debug_read(256).astype(np.int64)
a, b = d[:128].reshape(16, 8), d[128:].reshape(16, 8)
for p in range(16):
    t = b[p]
    print("p=%2d stage-in %4d | loads %4d | sweep %5d | publish %4d | mid %4d | loads %4d | sweep %5d | rest %4d | whole %5d"
          % (p, t[1]-t[0], t[2]-t[1], t[3]-t[2], t[4]-t[3], t[5]-t[4], t[6]-t[5], t[7]-t[6], a[p][3]-t[7], a[p][3]-a[p][2]))
ch.close()
