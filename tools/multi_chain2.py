import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
tot = 700
for C in (1, 2, 4):
    for mode in ("graph", "eager", "graph,no-overlap"):
        chains = [bnr_amd.Chain(X, y, 7, tot, 20240501, c + 1) for c in range(C)]
        for ch in chains:
            if mode == "eager": ch.set_option("graph", 0)
            if "no-overlap" in mode: ch.set_option("overlap", 0)
            ch.init_prior(); ch.run(2, tot, 50)
        t0 = time.time()
        for ch in chains: ch.run_async(51, tot, tot)
        for ch in chains: ch.sync()
        dt = time.time() - t0
        print("chains %d %-16s total %.0f it/s" % (C, mode, C * (tot - 50) / dt), flush=True)
        for ch in chains: ch.close()
