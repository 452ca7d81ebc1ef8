"""Diagnostic (-DBNR_STAMPS build, BNR_HIP_LIB=_stamps/libbnr_hip.so): phases of k_tail (s_memtime ticks of thread 0) for one chain: tools/stamps_tail.py [n V R]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
n, V, R = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (500, 300, 10)
X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=20240501)
ch = bnr_amd.Chain(X, y, R, 30, 21, 1)
ch.init_prior()
ch.set_option("graph", 0); ch.set_option("overlap", 0)          # alone on the chip
ch.run(2, 30, 30)
d = ch.debug_read(270).astype(np.int64)[256:264]
print("k_tail n=%d V=%d R=%d, alone on the chip, ticks between stamps 0..6:" % (n, V, R), " | ".join(str(int(d[i + 1] - d[i])) for i in range(6)), "| total", int(d[6] - d[0]))
ch.close()
