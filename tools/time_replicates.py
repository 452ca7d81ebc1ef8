import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import bnr_amd
from replicates import window_stats
d = np.load("tests/golden/test1_xy.npz"); X, y = d["X"], d["y"]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
t = time.time()
chains = [bnr_amd.Chain(X, y, 5, 400, 7000, 1)]
chains += [bnr_amd.Chain.like(chains[0], 7000 + i, 1, 400) for i in range(1, N)]
print("create", time.time() - t); t = time.time()
for c in chains: c.init_prior()
print("init", time.time() - t); t = time.time()
grp = bnr_amd.Group(chains)
grp.run(2, 200, 400)
print("run", time.time() - t); t = time.time()
tabs = [c.fetch() for c in chains]
print("fetch", time.time() - t); t = time.time()
reps = np.array([window_stats(tb, 200, 200) for tb in tabs])
print("stats", time.time() - t)
