import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
tot = 300
ch = bnr_amd.Chain(X, y, 7, tot, 20240501, 1)
ch.init_prior()
first = 2
while first <= tot:
    last = min(tot, first + 23)
    try:
        ch.run(first, tot, last)
    except Exception as e:
        print("FAILED rows", first, last, e)
        t = ch.fetch(max(1, first - 1), last)
        bad = np.isnan(t["gamma"]).any(axis=(1, 2))
        print("first nan gamma row", max(1, first - 1) + int(np.argmax(bad)), "tau2", t["tau2"][:8, 0, 0])
        break
    first = last + 1
else:
    print("ok", ch.counters())
