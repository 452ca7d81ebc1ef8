"""Every kernel of the sweep alone on the chip (options graph = 0, overlap = 0: one stream, eager launches) at the headline shape, 8 chains or one:
tools/serial_kernels.py [chains] -- run under rocprofv3 --kernel-trace --stats (tools/prof_cmd.sh) and compare with the two-branch schedule's durations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
C = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, V, R = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (500, 100, 7)
X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=20240501)
tot = 120
chains = [bnr_amd.Chain(X, y, R, tot, 20240501, 1)]
chains += [bnr_amd.Chain.like(chains[0], 20240501, c, tot) for c in range(2, C + 1)]
for ch in chains: ch.init_prior()
r = bnr_amd.Group(chains) if C > 1 else chains[0]
r.set_option("graph", 0); r.set_option("overlap", 0)
r.run(2, tot, tot)
print("serial", C, "chains", chains[0].counters())
