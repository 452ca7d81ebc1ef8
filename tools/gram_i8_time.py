#!/usr/bin/env python3
"""profiles/round5_gram_i8.txt: the Gram of a 0/1 model matrix on the i8 matrix pipe (k_sdigits + k_gram_i8) beside the f64 Gram (k_gram / k_gram8)
of the same chains on the same data -- HIP events around the launch (eager, single stream), and time per sweep from replayed graphs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
print("# shape, chains | Gram launch us (events, single stream): i8 (digits + Gram) / f64 | us per sweep of all chains (graph replay): i8 / f64 | it/s i8 / f64")
for (n, V, R, nch, tot) in ((500, 100, 7, 8, 160), (500, 100, 7, 1, 160), (500, 300, 10, 8, 72), (500, 300, 10, 1, 112), (200, 50, 5, 1, 248), (2000, 200, 7, 1, 64), (100, 30, 5, 2, 248)):
    rng = np.random.default_rng(7)
    q = V * (V + 1) // 2
    Xb = np.asfortranarray(rng.random((n, q)) < 0.5)
    y = rng.normal(size=n)
    res = {}
    for mode in ("i8", "f64"):
        ch = bnr_amd.Chain(bnr_amd.XInput(Xb, False), y, R, tot, 5, 1)
        members = [ch] + [bnr_amd.Chain.like(ch, 5, c, tot) for c in range(2, nch + 1)]
        for c in members:
            c.set_option("gram_i8", 1 if mode == "i8" else 0)
            c.init_prior()
        runner = bnr_amd.Group(members) if nch > 1 else ch
        L = ch.last_timing(4)[1]
        runner.set_option("overlap", 0); runner.set_profiling(True)
        runner.run(2, tot, 40)
        gus, ng = runner.last_timing(1)
        runner.set_profiling(False); runner.set_option("overlap", 1)
        runner.run(41, tot, 48)                      # capture + warm up
        bnr_amd.device_synchronize()
        t0 = time.perf_counter()
        runner.run(49, tot, tot)
        bnr_amd.device_synchronize()
        dt = time.perf_counter() - t0
        res[mode] = (gus, dt / (tot - 48) * 1e6, L)
        assert ch.counters()["chol_fail"] == 0
        if nch > 1: runner.close()
        for c in members: c.close()
    print("n=%d V=%d q=%d R=%d, %d chain%s (i8L = %d) | %.1f / %.1f | %.1f / %.1f | %.0f / %.0f" % (n, V, q, R, nch, "s" if nch > 1 else "", res["i8"][2], res["i8"][0], res["f64"][0],
          res["i8"][1], res["f64"][1], nch * 1e6 / res["i8"][1], nch * 1e6 / res["f64"][1]), flush=True)
