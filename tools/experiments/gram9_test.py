import sys, os
os.environ.setdefault("BNR_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesiannetworkregression.jl_amd", "csrc", "_var", "exp.so"))   # the experiments build (tools/r4_build_variants.sh)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
n, V, R = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (70, 19, 5)
X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=7)
ch = bnr_amd.Chain(X, y, R, 6, 3, 1)
ch.init_prior()
ch.run(2, 6, 3)
for v in (8, 9):
    ch.set_option("gram_variant", v)
    print("variant", v, flush=True)
    print("  gram us:", ch.debug_time_gram(3), flush=True)
    g = ch.debug_copy(3, 64 * 64 * 2)
    print("  partial tile checksum", float(np.sum(g)), float(g[5]), flush=True)
    d = ch.debug_read(620)
    print("  dbg", d[600:612], flush=True)
