"""Diagnostic (-DBNR_STAMPS build, BNR_HIP_LIB=_stamps/libbnr_hip.so): k_xpass_group's workgroups on the 100 MHz clock at the headline shape, 8 chains --
entry | W and sqrt(S) z1 staged | columns read and multiplied | partial vectors stored; alone on the chip (graph = 0, overlap = 0) and in the two-branch schedule."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, bnr_amd
n, V, R, C, tot = 500, 100, 7, 8, 30
X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=20240501)
for serial in (1, 0):
    chains = [bnr_amd.Chain(X, y, R, tot, 21, 1)]
    chains += [bnr_amd.Chain.like(chains[0], 21, c, tot) for c in range(2, C + 1)]
    for ch in chains: ch.init_prior()
    r = bnr_amd.Group(chains)
    if serial:
        r.set_option("graph", 0); r.set_option("overlap", 0)
    r.run(2, tot, tot)
    nb = 158 * 2
    w = chains[0].debug_read(400 + 4 * nb).astype(np.int64)[400:].reshape(nb, 4)
    t0 = w[:, 0].min()
    u = (w - t0) / 100.0
    dec = lambda a: " ".join("%5.1f" % np.sort(a)[int(i * (len(a) - 1) / 10)] for i in range(11))
    print("k_xpass_group, %d workgroups, %s (us after the first one starts; deciles over the workgroups)" % (nb, "alone on the chip" if serial else "two-branch schedule, graph replay"))
    print("   entry          ", dec(u[:, 0]))
    print("   staged         ", dec(u[:, 1]))
    print("   columns done   ", dec(u[:, 2]))
    print("   stored         ", dec(u[:, 3]))
    print("   per workgroup: staging %s | columns %s | stores %s" % (dec(u[:, 1] - u[:, 0]), dec(u[:, 2] - u[:, 1]), dec(u[:, 3] - u[:, 2])))
    r.close()
    for ch in chains: ch.close()
