"""Bitwise check of the linear schedule (option linear = 1 / 2 / 4) against the default schedule: a lockstep group of 8 chains, 70 sweeps."""
import sys, os
os.environ.setdefault("BNR_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesiannetworkregression.jl_amd", "csrc", "_var", "exp.so"))   # the experiments build (tools/r4_build_variants.sh)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
n, V, R = (int(v) for v in os.environ.get("BNR_SHAPE", "500,100,7").split(","))
X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=20240501)
tot = 72
def run(opts):
    ch = bnr_amd.Chain(X, y, R, tot, 5, 1)
    members = [ch] + [bnr_amd.Chain.like(ch, 5, c, tot) for c in range(2, 9)]
    for c in members: c.init_prior()
    g = bnr_amd.Group(members)
    for k, v in opts.items(): g.set_option(k, v)
    g.run(2, tot, 37)
    g.run(38, tot, tot)
    tabs = [c.fetch() for c in members]
    cnt = members[0].counters()
    g.close()
    for c in members: c.close()
    return tabs, cnt
ref, _ = run({})
for spec in sys.argv[1:]:
    opts = {k: int(v) for k, v in (kv.split("=") for kv in spec.split(","))}
    try:
        tabs, cnt = run(opts)
    except Exception as e:
        print(spec, "FAILED:", e); continue
    bad = [(i, k) for i, (a, b) in enumerate(zip(ref, tabs)) for k in a if not np.array_equal(a[k], b[k], equal_nan=True)]
    print(spec, "bitwise equal" if not bad else "DIFFERS in %d (chain, column) pairs, first %s" % (len(bad), bad[:4]), cnt)
