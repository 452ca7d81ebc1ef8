"""Diagnostic (-DBNR_STAMPS build): how the workgroups of one k_chol_step launch of a lockstep group spread in time.  Per panel step, relative
to the start of the first panel workgroup of the launch (s_memrealtime, 10 ns units): when the LAST panel workgroup starts, when the last
panel workgroup ends, when the last update workgroup ends."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
for nb in (1, 8):
    chains = [bnr_amd.Chain(X, y, 7, 40, 20240501, 1)]
    chains += [bnr_amd.Chain.like(chains[0], 20240501, c, 40) for c in range(2, nb + 1)]
    for ch in chains: ch.init_prior()
    r = bnr_amd.Group(chains) if nb > 1 else chains[0]
    for kv in sys.argv[1:]:
        k, v = kv.split("="); r.set_option(k, int(v))
    r.run(2, 40, 40)
    d = np.stack([ch.debug_read(16 * 8).reshape(16, 8).astype(np.int64) for ch in chains])      # chain, step, slot
    print("chains in the launch:", nb)
    prev_end = None
    for p in range(16):
        t0 = d[:, p, 1].min()
        endA, endB = d[:, p, 5].max(), d[:, p, 7].max() if p else 0
        print("  p=%2d first WG0 start 0 | other chains' WG0 start +%.2f | last panel WG start +%.2f | WG0 of chain 1 busy %.2f us | last panel WG end +%.2f | last update WG end +%.2f | since previous launch's last end %s"
              % (p, (d[:, p, 1].max() - t0) / 100, (d[:, p, 6].max() - t0) / 100, (d[0, p, 4] - d[0, p, 0]) / 2400.0, (endA - t0) / 100, (endB - t0) / 100 if p else 0.0,
                 "-" if prev_end is None else "%.2f" % ((t0 - prev_end) / 100)))
        prev_end = max(endA, endB)
    if nb > 1: r.close()
    for ch in chains: ch.close()
