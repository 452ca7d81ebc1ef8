"""Diagnostic (-DBNR_STAMPS build): every block of k_backproj on the 100 MHz clock -- start, dot products done, end: tools/experiments/stamps_bp_cfg4.py [n V R]  (default: BASELINE configs[3])"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, bnr_amd
n, V, R = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2000, 200, 7)
tot = 12
X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=20240501)
ch = bnr_amd.Chain(X, y, R, tot, 21, 1)
ch.init_prior(); ch.set_option("split_sums", 0)
ch.run(2, tot, tot)
nb = min((V * (V + 1) // 2 + 31) // 32, 1000)
w = ch.debug_read(400 + 2 * nb).astype(np.int64)[400:].reshape(nb, 2)
mid = ch.debug_read(2000 + nb).astype(np.int64)[2000:]
t0 = w[:, 0].min()
st, en, md = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0, (mid - t0) / 100.0
dec = lambda a: " ".join("%6.1f" % np.sort(a)[int(i * (len(a) - 1) / 10)] for i in range(11))
print("k_backproj n=%d V=%d R=%d: %d blocks (us after the first start; deciles)" % (n, V, R, nb))
print("   start     ", dec(st)); print("   dots done ", dec(md)); print("   end       ", dec(en))
print("   dots phase", dec(md - st)); print("   draws+sums", dec(en - md))
ch.close()
