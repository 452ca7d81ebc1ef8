"""Diagnostic (-DBNR_STAMPS build, BNR_HIP_LIB=_stamps/libbnr_hip.so): k_backproj3's workgroups (one per CU) on the 100 MHz clock -- when streaming ends, when every
item's draws and sums are done: tools/stamps_bp3.py <cfg5 | cfg5b | head8>  (config 5 real / Bool X, one chain; headline shape, 8 chains)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, bnr_amd
what = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
if what == "head8":
    n, V, R, C = 500, 100, 7, 8
else:
    n, V, R, C = 500, 300, 10, 1
tot = 30
rng = np.random.default_rng(9)
if what == "cfg5b":
    X = bnr_amd.XInput(np.asfortranarray(rng.random((n, V * (V + 1) // 2)) < 0.5), False)
    y = rng.normal(size=n)
else:
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=20240501)
chains = [bnr_amd.Chain(X, y, R, tot, 21, 1)]
chains += [bnr_amd.Chain.like(chains[0], 21, c, tot) for c in range(2, C + 1)]
for ch in chains:
    if what == "cfg5b":
        ch.set_option("gram_i8", 0)
    ch.init_prior()
r = bnr_amd.Group(chains) if C > 1 else chains[0]
r.set_option("cu_backproj", 1)          # (a build with tools/experiments/backproj_pipelines.patch)
r.run(2, tot, tot)
w = chains[0].debug_read(400 + 12 * 256).astype(np.int64)[400:].reshape(256, 12)
t0 = w[:, 0].min()
u = (w - t0) / 100.0
dec = lambda a: " ".join("%5.1f" % np.sort(a)[int(i * (len(a) - 1) / 10)] for i in range(11))
print(what, "k_backproj3, 256 workgroups, us after the first one starts (deciles over the workgroups)")
print("   start           ", dec(u[:, 0]))
print("   streaming done  ", dec(u[:, 1]))
for k in range(5):
    ok = w[:, 2 + 2 * k] > 0
    if ok.any():
        print("   item %d (%3d wgs): draws done" % (k, ok.sum()), dec(u[ok, 2 + 2 * k]))
        print("                      sums done ", dec(u[ok, 3 + 2 * k]))
