"""Diagnostic: per-task stamps of three persistent Gram workgroups (build with -DBNR_STAMPS)."""
import sys, os
os.environ.setdefault("BNR_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesiannetworkregression.jl_amd", "csrc", "_var", "stamps.so"))   # the experiments build (tools/r4_build_variants.sh)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
chains = [bnr_amd.Chain(X, y, 7, 8, 20240501, 1)]
chains += [bnr_amd.Chain.like(chains[0], 20240501, c, 8) for c in range(2, 9)]
for ch in chains: ch.init_prior()
g = bnr_amd.Group(chains)
for k, v in (("graph", 0), ("gram_variant", 9), ("overlap", 0), ("pipeline", 0)): g.set_option(k, v)
g.run(2, 8, 8)
d = chains[0].debug_read(600).astype(np.int64)
t0 = d[439]
for w in range(3):
    print("workgroup", (5, 300, 700)[w])
    for k in range(5):
        o = 440 + 40 * w + 8 * k
        if d[o] == 0: break
        print("   task id %5d queue %d: start %+8.2f us, compute+stores issued %7.2f us, published after %6.2f us" % (d[o + 4] & 0xffffff, d[o + 4] >> 24, (d[o] - t0) / 100.0, (d[o + 1] - d[o]) / 100.0, (d[o + 2] - d[o + 1]) / 100.0))
d = chains[0].debug_read(1000).astype(np.uint64)
print("first / second task duration (us) of every 6th workgroup: blockIdx xcc:se:cu queue")
rows = []
for i in range(128):
    a, b = int(d[640 + 2 * i]), int(d[640 + 2 * i + 1])
    if a == 0: continue
    meta = a >> 32
    rows.append((meta & 15, (meta >> 4) & 3, (meta >> 8) & 15, 6 * i, (a & 0xffffffff) / 100.0, (b & 0xffffffff) / 100.0, (meta >> 12) & 15))
rows.sort()
for r in rows: print("  xcc %d se %d cu %d  wg %3d  first %6.1f second %6.1f  queue %d" % r)
d = chains[0].debug_read(440).astype(np.int64)
print("tasks executed in all launches:", d[430], "= per launch", d[430] / 7.0, " id checksum", d[431])
