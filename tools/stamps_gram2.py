import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
ch = bnr_amd.Chain(X, y, 7, 8, 20240501, 1)
ch.init_prior(); ch.run(2, 8, 4)
for g in (1, 8, 32, 64, 126, 200, 240, 252):
    print("grid", g, "-> %.1f us per launch" % ch.debug_time_gram(g * 1000000 + 20))
