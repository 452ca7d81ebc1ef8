"""Diagnostic (-DBNR_STAMPS build): shader cycles per phase of every k_chol_step launch (panel workgroup 0 of chain 1) at any shape: stamps_steps.py n V R [chains] [option=value ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
opts = [a for a in sys.argv[1:] if "=" in a]
argv = [a for a in sys.argv[1:] if "=" not in a]
n, V, R = int(argv[0]), int(argv[1]), int(argv[2])
nb = int(argv[3]) if len(argv) > 3 else 1
X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=20240501)
chains = [bnr_amd.Chain(X, y, R, 40, 20240501, 1)]
chains += [bnr_amd.Chain.like(chains[0], 20240501, c, 40) for c in range(2, nb + 1)]
for ch in chains: ch.init_prior()
r = bnr_amd.Group(chains) if nb > 1 else chains[0]
for kv in opts:
    k, v = kv.split("="); r.set_option(k, int(v))
r.run(2, 40, 40)
nbk = ((n + 63) // 64 * 64) // 32          # n_pad is a multiple of the Gram tile (64)
d = chains[0].debug_read(nbk * 8).reshape(nbk, 8).astype(np.int64)
print("n=%d V=%d R=%d, %d chain(s): cycles of panel workgroup 0 of chain 1 [fetch + pending update | sweep | store], wave 0's total; the workgroup's last wave on the shader clock and on the 100 MHz clock" % (n, V, R, nb))
e = chains[0].debug_read(4096)[3900:3900 + 2 * nbk].reshape(nbk, 2).astype(np.int64)       # the last wave's end on both clocks
f = chains[0].debug_read(4096)[3800:3800 + 4 * nbk].reshape(nbk, 4).astype(np.int64)     # wave 0: kernel entry | loads back | MFMAs done
for p in range(nbk):
    t = d[p]
    if p: print("        wave 0 from the kernel's entry: role A starts %5d | loads back %5d | MFMAs done %5d | staged + barrier + own pivots + end %5d" % (t[0] - f[p, 0], f[p, 1] - f[p, 0], f[p, 2] - f[p, 0], t[4] - f[p, 0]))
    us = (e[p, 0] - t[1]) / 100.0
    print("  p=%2d  %6d %6d %6d   wave 0: %6d   last wave: %6d cycles in %.2f us = %.2f GHz" % (p, t[2] - t[0], t[3] - t[2], t[4] - t[3], t[4] - t[0], e[p, 1] - t[0], us, (e[p, 1] - t[0]) / us / 1e3))
if nb > 1: r.close()
for ch in chains: ch.close()
