"""Diagnostic (-DBNR_STAMPS build): phases inside bnr_panel_sweep_pipe of panel workgroup 0 (shader cycles since the function's entry): barrier | wave 0 done |
wave 1 starts / ends its own columns | wave 2 done | wave 3 starts / ends its own columns; and wave 0 from the kernel's role-A entry to its own end (NOT the workgroup's: the last wave ends ~4 500 cycles later, tools/stamps_steps.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
n, V, R = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (500, 100, 7)
X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=20240501)
ch = bnr_amd.Chain(X, y, R, 40, 20240501, 1)
ch.init_prior()
ch.run(2, 40, 40)
d = ch.debug_read(256).astype(np.int64)
a, b = d[:128].reshape(16, 8), d[128:].reshape(16, 8)
for p in range(min(16, (n + 63) // 64 * 2)):
    t = b[p]
    print("p=%2d barrier %4d | w0 done %5d | w1 own %5d..%5d | w2 done %5d | w3 own %5d..%5d | fetch + update %5d | wave 0 from entry to its end %5d"
          % (p, t[1]-t[0], t[2]-t[0], t[3]-t[0], t[4]-t[0], t[5]-t[0], t[6]-t[0], t[7]-t[0], a[p][2]-a[p][0], a[p][4]-a[p][0]))
ch.close()
