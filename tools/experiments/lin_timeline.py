"""Timeline of the linear schedule from the gates' and setters' own clocks (option linear_debug): per sweep and part, relative to the ring gate of part 0 (us):
S gate in/out (scalar branch may start), rhs done | ring gate in/out (Gram start), Gram done, solve gate in/out (factorization done / rhs there), back-projection done."""
import sys, os
os.environ.setdefault("BNR_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesiannetworkregression.jl_amd", "csrc", "_var", "exp.so"))   # the experiments build (tools/r4_build_variants.sh)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
P = int(sys.argv[1]); extra = sys.argv[2:]
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
tot = 400
ch = bnr_amd.Chain(X, y, 7, tot, 5, 1)
members = [ch] + [bnr_amd.Chain.like(ch, 5, c, tot) for c in range(2, 9)]
for c in members: c.init_prior()
g = bnr_amd.Group(members)
g.set_option("linear_debug", 1)
for kv in extra:
    k, v = kv.split("="); g.set_option(k, int(v))
g.set_option("linear", P)
g.run(2, tot, 200)
import time
t = time.time(); g.run(201, tot, tot - 8); dt = time.time() - t
print("%.1f us per sweep" % (1e6 * dt / (tot - 8 - 200)))
d = ch.debug_read(1024).astype(np.int64)[:16 * 4 * 12].reshape(16, 4, 6, 2)
n_last = (tot - 8 - 1) - 1            # sweeps are counted from 0 at the start of every run call: the last one of the second call
order = [k for k in range(1, 8)]
t0 = d[order[0], 0, 2, 1] if P > 1 else d[order[0], 0, 4, 0]
for n in order:
    for p in range(P):
        r = lambda kind, w: (d[n, p, kind, w] - t0) / 100.0
        merged = any(kv.startswith("linear_merge=1") for kv in extra)
        if merged:
            print("sweep %2d part %d | ring gate %8.1f -> %8.1f (Gram starts), Gram done %8.1f, back-projection done %8.1f" % (n, p, r(2, 0), r(2, 1), r(3, 0), r(5, 0)))
        else:
            print("sweep %2d part %d | S gate %8.1f -> %8.1f, rhs done %8.1f | ring gate %8.1f -> %8.1f (Gram starts), Gram done %8.1f, solve gate (factorization done) %8.1f -> %8.1f, back-projection done %8.1f"
                  % (n, p, r(0, 0), r(0, 1), r(1, 0), r(2, 0), r(2, 1), r(3, 0), r(4, 0), r(4, 1), r(5, 0)))
g.close()
for c in members: c.close()
