"""SHA-256 of the tables a few fixed runs produce (shapes incl. the headline group of 8 and BASELINE configs[4]'s size, real and Bool X): run it with two builds of the library
(BNR_HIP_LIB) and diff the output to see whether a change is bitwise neutral."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
def digest(tabs):
    h = hashlib.sha256()
    for t in tabs:
        for k in sorted(t):
            h.update(np.ascontiguousarray(t[k]).tobytes())
    return h.hexdigest()[:16]
for (n, V, R, C, tot, binary) in [(500, 100, 7, 8, 7, False), (500, 100, 7, 1, 7, False), (500, 300, 10, 1, 4, False), (500, 300, 10, 2, 4, True), (70, 19, 5, 3, 30, False), (200, 50, 5, 1, 12, False), (1000, 12, 3, 2, 5, False)]:
    if binary:
        rng = np.random.default_rng(5)
        X = bnr_amd.XInput(np.asfortranarray(rng.random((n, V * (V + 1) // 2)) < 0.5), False); y = rng.normal(size=n)
    else:
        X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=77)
    chains = [bnr_amd.Chain(X, y, R, tot, 9, 1)]
    chains += [bnr_amd.Chain.like(chains[0], 9, c, tot) for c in range(2, C + 1)]
    for ch in chains: ch.init_prior()
    r = bnr_amd.Group(chains) if C > 1 else chains[0]
    r.run(2, tot, tot)
    print("n=%d V=%d R=%d chains=%d rows=%d binary=%s: %s" % (n, V, R, C, tot, binary, digest([ch.fetch() for ch in chains])), flush=True)
    if C > 1: r.close()
    for ch in chains: ch.close()
