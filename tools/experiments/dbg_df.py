import sys, os
os.environ.setdefault("BNR_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesiannetworkregression.jl_amd", "csrc", "_var", "exp.so"))   # tools/r4_build_variants.sh "exp:-DBNR_EXPERIMENTS"
sys.path.insert(0, "/root/repo")
import numpy as np, bnr_amd
n, V, R = 500, 100, 7
X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=7)
def run(variant, nb, rows=2, gv=0):
    ch = bnr_amd.Chain(X, y, R, rows, 3, 1)
    mates = [bnr_amd.Chain.like(ch, 3, c, rows) for c in range(2, nb + 1)]
    for c in [ch] + mates: c.init_prior()
    g = bnr_amd.Group([ch] + mates) if nb > 1 else ch
    g.set_option("factor_variant", variant); g.set_option("graph", 0); g.set_option("overlap", 0); g.set_option("gram_variant", gv)
    g.run(2, rows, rows)
    np_ = 512; ldE = 2 * np_ + 32
    Es = [c.debug_copy(0, ldE * np_).reshape(np_, ldE).T.copy() for c in [ch] + mates]
    tabs = [c.fetch() for c in [ch] + mates]
    print("counters", ch.counters())
    return Es, tabs
ref = None
for nb in (1, 2):
    E0, t0 = run(0, nb)
    E4, t4 = run(4, nb)
    if ref is None: ref = (E0[0], E4[0])
    else:
        print("variant 0: group member 0 E == solo E on L and Y blocks:", all(np.array_equal(ref[0][32*r:32*r+32, 32*c:32*c+32], E0[0][32*r:32*r+32, 32*c:32*c+32]) for c in range(16) for r in range(c+1, 16)), all(np.array_equal(ref[0][512+32*r:512+32*r+32, 32*c:32*c+32], E0[0][512+32*r:512+32*r+32, 32*c:32*c+32]) for c in range(16) for r in range(0, c+1)))
        print("variant 4: group member 0 E == solo E on L and Y blocks:", all(np.array_equal(ref[1][32*r:32*r+32, 32*c:32*c+32], E4[0][32*r:32*r+32, 32*c:32*c+32]) for c in range(16) for r in range(c+1, 16)), all(np.array_equal(ref[1][512+32*r:512+32*r+32, 32*c:32*c+32], E4[0][512+32*r:512+32*r+32, 32*c:32*c+32]) for c in range(16) for r in range(0, c+1)))
        A, B = ref[1], E4[0]
        for (r, c) in ((1, 2), (2, 2), (0, 2), (1, 1)):
            a, b = A[512+32*r:512+32*r+32, 32*c:32*c+32], B[512+32*r:512+32*r+32, 32*c:32*c+32]
            print("Y", r, c, "max |solo - group| of variant 4:", np.max(np.abs(a - b)), "nonfinite", (~np.isfinite(b)).sum(), "first rows differing:", np.unique(np.argwhere(a != b)[:, 0])[:8], "cols", np.unique(np.argwhere(a != b)[:, 1])[:8])
    for m in range(nb):
        A, B = E0[m], E4[m]
        # compare block-wise: L blocks (matrix rows r > c) and Y blocks (identity rows r <= c)
        badb = []
        for c in range(16):
            for r in range(c + 1, 16):
                if not np.array_equal(A[32*r:32*r+32, 32*c:32*c+32], B[32*r:32*r+32, 32*c:32*c+32]): badb.append(("L", r, c))
            for r in range(0, c + 1):
                if not np.array_equal(A[512+32*r:512+32*r+32, 32*c:32*c+32], B[512+32*r:512+32*r+32, 32*c:32*c+32]): badb.append(("Y", r, c))
        print("nb", nb, "member", m, "blocks that differ:", len(badb), badb[:12], "gamma equal:", np.array_equal(t0[m]["gamma"], t4[m]["gamma"]))


def sums(nb):
    ch = bnr_amd.Chain(X, y, R, 2, 3, 1)
    mates = [bnr_amd.Chain.like(ch, 3, c, 2) for c in range(2, nb + 1)]
    for c in [ch] + mates: c.init_prior()
    g = bnr_amd.Group([ch] + mates) if nb > 1 else ch
    g.set_option("factor_variant", 4); g.set_option("graph", 0); g.set_option("overlap", 0)
    g.run(2, 2, 2)
    d = ch.debug_read(4096).view(np.float64)[:2048].reshape(64, 4, 4, 2)
    return d
a, b = sums(1), sums(2)
for w in range(64):
    for k in range(4):
        if not np.array_equal(a[w, k], b[w, k]):
            slot, grp = w & 15, w >> 4
            j = grp + 4 * k
            print("w", w, "slot", slot, "grp", grp, "k", k, "column", j, "D sums differ:", not np.array_equal(a[w, k, :, 0], b[w, k, :, 0]), "B sums differ:", not np.array_equal(a[w, k, :, 1], b[w, k, :, 1]), "waves", [i for i in range(4) if not np.array_equal(a[w, k, i], b[w, k, i])])
