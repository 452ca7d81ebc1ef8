"""Diagnostic: what the workgroups of the resident Gram k_gram8q do with their time (-DBNR_STAMPS build): tools/stamps_g8q.py [nchains] [resv_mask]"""
import sys, os
os.environ.setdefault("BNR_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesiannetworkregression.jl_amd", "csrc", "_var", "stamps.so"))   # the experiments build (tools/r4_build_variants.sh)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mask = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 13
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
chains = [bnr_amd.Chain(X, y, 7, 8, 20240501, 1)]
chains += [bnr_amd.Chain.like(chains[0], 20240501, c, 8) for c in range(2, nb + 1)]
for ch in chains: ch.init_prior()
g = bnr_amd.Group(chains) if nb > 1 else chains[0]
for k, v in (("graph", 0), ("gram_variant", variant), ("overlap", 0), ("resv_mask", mask)): g.set_option(k, v)
g.run(2, 8, 8)
d = chains[0].debug_read(1000).astype(np.uint64)
rows = []
for i in range(66, 250):
    t0, t1, ft, meta = (int(v) for v in d[4 * i:4 * i + 4])
    if t0 == 0: continue
    rows.append(dict(t0=t0, t1=t1, desc=(ft & 0xfffff) / 100.0, task=((ft >> 20) & 0xfffff) / 100.0, ticket=((ft >> 40) & 0xfffff) / 100.0, nt=meta & 0xffff, xcc=(meta >> 32) & 15, cu=(meta >> 36) & 15, se=(meta >> 40) & 7, wg=3 * i))
tmin = min(r["t0"] for r in rows)
print("variant %d mask 0x%x: %d workgroups stamped; launch span %.1f us" % (variant, mask, len(rows), (max(r["t1"] for r in rows) - tmin) / 100.0))
print("  wg   xcc se cu   start    end   tasks  ticket  descriptors  in tasks (us)")
for r in rows[::8]:
    print("  %4d  %d  %d  %2d  %6.1f %6.1f   %3d  %6.2f   %6.2f   %7.2f" % (r["wg"], r["xcc"], r["se"], r["cu"], (r["t0"] - tmin) / 100.0, (r["t1"] - tmin) / 100.0, r["nt"], r["ticket"], r["desc"], r["task"]))
for k in ("ticket", "desc", "task"):
    v = np.array([r[k] for r in rows])
    print("%-7s per workgroup: mean %.2f us, min %.2f, max %.2f" % (k, v.mean(), v.min(), v.max()))
nt = np.array([r["nt"] for r in rows])
print("tasks per workgroup: min %d mean %.2f max %d; per task: descriptors %.2f us, task %.2f us" % (nt.min(), nt.mean(), nt.max(), sum(r["desc"] for r in rows) / nt.sum(), sum(r["task"] for r in rows) / nt.sum()))
print("start times (us):", np.percentile([(r["t0"] - tmin) / 100.0 for r in rows], [0, 50, 100]).round(1), " end times (us):", np.percentile([(r["t1"] - tmin) / 100.0 for r in rows], [0, 10, 50, 90, 100]).round(1))
