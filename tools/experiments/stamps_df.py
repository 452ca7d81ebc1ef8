"""Diagnostic (-DBNR_STAMPS build): the data-flow factorization's steps from inside -- for the slot-0 sweeper of every column q: when it entered
iteration q, started / ended its sweep, published, and when it had applied panel q-1 (tools/stamps_df.py [nchains])."""
import sys, os
os.environ.setdefault("BNR_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesiannetworkregression.jl_amd", "csrc", "_var", "stamps.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
chains = [bnr_amd.Chain(X, y, 7, 8, 20240501, 1)]
chains += [bnr_amd.Chain.like(chains[0], 20240501, c, 8) for c in range(2, nb + 1)]
for ch in chains: ch.init_prior()
g = bnr_amd.Group(chains) if nb > 1 else chains[0]
for k, v in (("graph", 0), ("factor_variant", 4), ("overlap", 0)): g.set_option(k, v)
g.run(2, 8, 8)
d = chains[0].debug_read(1000).astype(np.int64)
t0 = d[400 + 1]
print("q   enter   sweep start  sweep end  published | panel q-1 applied (all times us after the first sweep started)")
prev_pub = None
for q in range(16):
    a = d[400 + 8 * q: 400 + 8 * q + 5]
    f = lambda v: (v - t0) / 100.0
    print("%2d %8.2f %10.2f %10.2f %10.2f | %8.2f    sweep %.2f us, publish %.2f, from the previous column's publish to this sweep's start %.2f" % (
        q, f(a[0]), f(a[1]), f(a[2]), f(a[3]), f(a[4]) if a[4] else float("nan"), (a[2] - a[1]) / 100.0, (a[3] - a[2]) / 100.0, (a[1] - prev_pub) / 100.0 if prev_pub else float("nan")))
    prev_pub = a[3]
print(chains[0].counters())
