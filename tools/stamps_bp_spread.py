"""Diagnostic (-DBNR_STAMPS build): when every workgroup of the back-projection of the LAST sweep enters and leaves (s_memrealtime), per chain of a lockstep group:
stamps_bp_spread.py [chains]  -- the entry times tell whether the launch is resident at once or runs in rounds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
chains = [bnr_amd.Chain(X, y, 7, 40, 20240501, 1)]
chains += [bnr_amd.Chain.like(chains[0], 20240501, c, 40) for c in range(2, nb + 1)]
for ch in chains: ch.init_prior()
r = bnr_amd.Group(chains) if nb > 1 else chains[0]
r.run(2, 40, 40)
nblk = (5050 + 31) // 32
d = np.stack([ch.debug_read(400 + 2 * nblk)[400:].reshape(nblk, 2).astype(np.int64) for ch in chains])     # chain, block, (in, out)
t0 = d[:, :, 0].min()
tin, tout = (d[:, :, 0] - t0) / 100.0, (d[:, :, 1] - t0) / 100.0
print("%d chain(s): %d workgroups; entries (us after the first): deciles %s" % (nb, tin.size, np.round(np.percentile(tin, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100]), 2)))
print("   exits: deciles %s" % np.round(np.percentile(tout, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100]), 2))
print("   time inside a workgroup: deciles %s" % np.round(np.percentile(tout - tin, [0, 10, 50, 90, 100]), 2))
late = tin > 2.0
print("   workgroups entering more than 2 us after the first: %d (blocks %s ... of chains %s)" % (late.sum(), np.unique(np.where(late)[1])[:12], np.unique(np.where(late)[0])))
if nb > 1: r.close()
for ch in chains: ch.close()
