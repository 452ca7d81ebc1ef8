import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
tot = 3000
for mode in ("eager+overlap", "eager+overlap,chunk200", "graph+overlap,chunk200", "graph+overlap,chunk200", "graph+overlap,chunk200"):
    ch = bnr_amd.Chain(X, y, 7, tot, 20240501, 1)
    if mode.startswith("eager"): ch.set_option("graph", 0)
    if "no-overlap" in mode: ch.set_option("overlap", 0)
    ch.init_prior()
    first = 2
    ok = True
    while first <= tot and ok:
        last = min(tot, first + (199 if 'chunk200' in mode else 49))
        try:
            ch.run(first, tot, last)
        except Exception as e:
            print(mode, "FAILED in rows", first, last, e, ch.counters())
            t = ch.fetch(max(1, first - 2), last)
            print("first nan gamma row", max(1, first - 2) + int(np.argmax(np.isnan(t["gamma"]).any(axis=(1, 2)))))
            ok = False
        first = last + 1
    if ok: print(mode, "ok", ch.counters())
    ch.close()
