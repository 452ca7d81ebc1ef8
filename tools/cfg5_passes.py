"""BASELINE configs[4]'s size (n=500, V=300, q=45 150), one chain: tools/cfg5_passes.py <x: real | bool8 | bool64> <back-projection kernel: 0 shipped | 1 k_backproj2 | 2 k_backproj3 (patch builds)> [chains]
real = real-valued model matrix (f64 image), bool8 = 0/1 matrix read through its byte image, bool64 = 0/1 matrix read through the f64 image.
Run under rocprofv3 --kernel-trace --stats (tools/cfg5_passes.sh) and compare k_xpass / k_backproj / k_backproj2."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
kind, pair = sys.argv[1], int(sys.argv[2])
C = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n, V, R, tot = 500, 300, 10, 60
rng = np.random.default_rng(9)
if kind == "real":
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=20240501)
else:
    X = bnr_amd.XInput(np.asfortranarray(rng.random((n, V * (V + 1) // 2)) < 0.5), False)
    y = rng.normal(size=n)
chains = [bnr_amd.Chain(X, y, R, tot, 21, 1)]
chains += [bnr_amd.Chain.like(chains[0], 21, c, tot) for c in range(2, C + 1)]
for ch in chains:
    if kind != "real":
        ch.set_option("byte_x", 1 if kind == "bool8" else 0)
        pass
    ch.init_prior()
runner = bnr_amd.Group(chains) if C > 1 else chains[0]
if pair:                                   # 1 = k_backproj2, 2 = k_backproj3: only in a build with tools/experiments/backproj_pipelines.patch applied
    runner.set_option("pair_backproj", pair & 1); runner.set_option("cu_backproj", pair >> 1)
runner.run(2, tot, tot)
print(kind, "pair_backproj", pair, "chains", C, "byte image in use:", chains[0].last_timing(3)[1], chains[0].counters())
