"""Diagnostic (-DBNR_STAMPS build): one sweep's kernels on ONE clock, untraced and under graph replay -- what rocprofv3 cannot show without slowing the sweep by a quarter.
Every kernel of the sweep stamps s_memrealtime (100 MHz, the same on every XCD) at the entry of its workgroup 0, the latest entry and the latest exit of any of its workgroups
(chain 1's debug words from 4000; the panel steps use their own words 8 p + 1 / 5 / 7); atomicMax keeps the LAST sweep's values.  Usage: stamps_timeline.py <chains> [n V R] [opt=val ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
args = [a for a in sys.argv[1:] if "=" not in a]
opts = [a for a in sys.argv[1:] if "=" in a]
nb = int(args[0]) if args else 1
n, V, R = (int(args[1]), int(args[2]), int(args[3])) if len(args) >= 4 else (500, 100, 7)
X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=20240501)
chains = [bnr_amd.Chain(X, y, R, 48, 20240501, 1)]
chains += [bnr_amd.Chain.like(chains[0], 20240501, c, 48) for c in range(2, nb + 1)]
for ch in chains: ch.init_prior()
r = bnr_amd.Group(chains) if nb > 1 else chains[0]
for kv in opts:
    k, v = kv.split("="); r.set_option(k, int(v))
r.run(2, 48, 48)
KINDS = ["gram", "solve_w", "solve_a4", "backproj", "psum", "tail", "tail_a", "node", "xpass", "rhs", "sdigits"]
d = np.stack([ch.debug_read(4096).astype(np.int64) for ch in chains])       # chain, word
nbk = ((n + 63) // 64 * 64) // 32          # n_pad is a multiple of the Gram tile (64)
rows = []
for k, name in enumerate(KINDS):
    w = d[:, 4000 + 4 * k: 4000 + 4 * k + 3]
    if w[:, 1].max() == 0: continue
    rows.append((w[:, 0].min() if w[:, 0].max() else w[:, 1].min(), w[:, 1].max(), w[:, 2].max(), name))
for p in range(nbk):
    w = d[:, 8 * p: 8 * p + 8]
    if w[:, 1].max() == 0: continue
    rows.append((w[:, 1].min(), w[:, 6].max(), max(w[:, 5].max(), w[:, 7].max()), "chol %2d" % p))
# the back-projection of the last sweep ends the sweep: everything relative to the Gram's start of that sweep (the Gram follows the previous back-projection)
t0 = [r_ for r_ in rows if r_[3] == "gram"][0][0]
print("%d chain(s), n=%d V=%d R=%d: last sweep of the run, us from the entry of the Gram's first workgroup" % (nb, n, V, R))
print("   first in   last in      out   (out - first in)")
for a, b, c, name in sorted(rows):
    print(" %9.2f %9.2f %9.2f   %7.2f  %s" % ((a - t0) / 100, (b - t0) / 100, (c - t0) / 100, (c - a) / 100, name))
if nb > 1: r.close()
for ch in chains: ch.close()
