"""Diagnostic (-DBNR_STAMPS build, BNR_HIP_LIB=_stamps/libbnr_hip.so): phases of k_backproj / k_backproj2 workgroup 3 (k_backproj: block 7) of a chain alone at
BASELINE configs[4]'s size: tools/stamps_bp2.py <real|bool8>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, bnr_amd
kind = sys.argv[1] if len(sys.argv) > 1 else "real"
n, V, R, tot = 500, 300, 10, 30
rng = np.random.default_rng(9)
if kind == "real":
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=20240501)
else:
    X = bnr_amd.XInput(np.asfortranarray(rng.random((n, V * (V + 1) // 2)) < 0.5), False)
    y = rng.normal(size=n)
for pair in (0, 1):
    ch = bnr_amd.Chain(X, y, R, tot, 21, 1)
    if kind != "real":
        ch.set_option("gram_i8", 0)
    ch.set_option("pair_backproj", pair)
    ch.init_prior()
    ch.run(2, tot, tot)
    full = ch.debug_read(360).astype(np.int64)
    if pair == 0:
        d = full[320:324]
        print("%s k_backproj block 7: staging + dots %d | GIG %d | sums %d | total %d ticks" % (kind, d[1]-d[0], d[2]-d[1], d[3]-d[2], d[3]-d[0]))
    else:
        p = full[340:350] - full[340]
        print("%s k_backproj2 workgroup 3 (ticks after entry): staged %d | dots A %d | wave 0 draws A done %d, wave 1 %d | waves 2 dots B done %d | barrier %d | draws B done %d | sums B %d, sums A %d" %
              (kind, p[1], p[2], p[3], p[9], p[4], p[5], p[6], p[7], p[8]))
        w = ch.debug_read(400 + 2 * 706).astype(np.int64)[400:].reshape(-1, 2)
        w = w[w[:, 1] > 0]
        t0 = w[:, 0].min()
        st, en = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0
        print("   %d workgroups (100 MHz clock): start min %.2f median %.2f max %.2f us | end min %.2f median %.2f max %.2f us | lifetime min %.2f median %.2f max %.2f us" %
              (len(w), st.min(), np.median(st), st.max(), en.min(), np.median(en), en.max(), (en - st).min(), np.median(en - st), (en - st).max()))
        ph = ch.debug_read(2000 + 2 * 706).astype(np.int64)[2000:].reshape(-1, 2)[:len(w)]
        p1, p2 = (ph[:, 0] - t0) / 100.0, (ph[:, 1] - t0) / 100.0
        dec = lambda a: " ".join("%.1f" % np.sort(a)[int(i * (len(a) - 1) / 10)] for i in range(11))
        print("   phase 1 ends by decile:", dec(p1))
        print("   phase 2 ends by decile:", dec(p2))
        print("   phase 3 + sums duration by decile:", dec(en - p2))
        slow = np.argsort(en)[-70:]
        print("   the slowest 70 workgroups: ids mod 8:", np.bincount(slow % 8, minlength=8), "ids (sorted):", np.sort(slow)[:40], "their phase-1 end median %.1f, phase-2 end median %.1f" % (np.median(p1[slow]), np.median(p2[slow])))
        order = np.argsort(st)
        print("   start by decile:", " ".join("%.1f" % st[order[int(i * (len(w) - 1) / 10)]] for i in range(11)))
        print("   end by decile:  ", " ".join("%.1f" % np.sort(en)[int(i * (len(w) - 1) / 10)] for i in range(11)))
    ch.close()
