"""Diagnostic (-DBNR_STAMPS build, BNR_HIP_LIB=_stamps/libbnr_hip.so): the shader clock of the PRODUCT Gram kernel's K loop.
k_gram8<bnr_many> stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) in front of and behind the K loop of every workgroup; after >= 2 s of back-to-back sweeps of the
bench's 8-chain group the last launch's stamps give, per workgroup, clock = d(memtime) / d(memrealtime) x 100 MHz.  Also: the loop's share of the launch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
n, V, R, C = 500, 100, 7, 8
X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=20240501)
K = 700
chains = [bnr_amd.Chain(X, y, R, K + 10, 20240501, 1)]
chains += [bnr_amd.Chain.like(chains[0], 20240501, c, K + 10) for c in range(2, C + 1)]
for ch in chains: ch.init_prior()
g = bnr_amd.Group(chains)
g.prepare()
t0 = time.perf_counter()
first = 2
reps = 0
while time.perf_counter() - t0 < 2.5:                     # >= 2 s of back-to-back sweeps (the table is a ring here: rows are simply overwritten run after run)
    g.run(2, K, K)
    reps += 1
bnr_amd.device_synchronize(0)
wall = time.perf_counter() - t0
clk, cyc, us = [], [], []
for ch in chains:
    d = ch.debug_read(2048 + 4 * 256).astype(np.int64)[2048:].reshape(256, 4)
    d = d[(d[:, 3] > d[:, 1]) & (d[:, 1] > 0)]
    clk += list((d[:, 2] - d[:, 0]) / ((d[:, 3] - d[:, 1]) * 10.0))       # cycles per 10 ns -> GHz
    cyc += list(d[:, 2] - d[:, 0]); us += list((d[:, 3] - d[:, 1]) / 100.0)
clk, cyc, us = np.array(clk), np.array(cyc), np.array(us)
print("%d sweeps of %d chains in %.2f s (%.1f us per sweep under the stamps build); K loops of the last launch: %d workgroups" % (reps * (K - 1), C, wall, 1e6 * wall / (reps * (K - 1)), len(clk)))
print("shader clock inside the K loop of k_gram8<bnr_many> (GHz): median %.3f  p10 %.3f  p90 %.3f  min %.3f  max %.3f" % (np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90), clk.min(), clk.max()))
print("K loop per workgroup: median %.1f us = %d cycles (23 batches of 8 columns x 2 K-groups: 8 x 64-cycle MFMAs per batch and wave = 11 776 cycles of pure MFMA issue per SIMD for 3 resident workgroups)" % (np.median(us), int(np.median(cyc))))
g.close()
for ch in chains: ch.close()
