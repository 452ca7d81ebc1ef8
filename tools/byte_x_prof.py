"""cfg5-size chain (n=500, V=300) on a 0/1 model matrix, the X passes reading the byte image (arg 1) or the f64 image (arg 0): run under
rocprofv3 --kernel-trace --stats and compare k_xpass / k_backproj."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
byte_x = int(sys.argv[1])
n, V, R, tot = 500, 300, 10, 60
rng = np.random.default_rng(9)
X = np.asfortranarray(rng.random((n, V * (V + 1) // 2)) < 0.5)
y = rng.normal(size=n)
ch = bnr_amd.Chain(bnr_amd.XInput(X, False), y, R, tot, 21, 1)
ch.set_option("byte_x", byte_x)
ch.init_prior()
ch.run(2, tot, tot)
print("byte image in use:", ch.last_timing(3)[1], ch.counters())
