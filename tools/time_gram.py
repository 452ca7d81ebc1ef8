import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
for (n, V, R) in [(500, 100, 7), (200, 50, 5), (2000, 200, 7), (500, 300, 10)]:
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=20240501)
    ch = bnr_amd.Chain(X, y, R, 8, 20240501, 1)
    ch.init_prior(); ch.run(2, 8, 4)
    us = ch.debug_time_gram(200)
    q = V * (V + 1) // 2
    print("n=%d V=%d q=%d: k_gram standalone %.1f us -> %.1f TFLOP/s algorithmic (n^2 q)" % (n, V, q, us, n * n * q / us / 1e6))
    ch.close()
