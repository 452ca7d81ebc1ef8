#!/usr/bin/env python3
"""Round 4, first measurement pass: sweep time of the 8-chain group (and of one chain) at the headline size under the new options, the Gram
kernels alone, and -- on a -DBNR_EXPERIMENTS build -- the critical chain without the scalar branch beside it.
  BNR_HIP_LIB=<build> tools/r4_exp1.py [sweeps]"""
import sys, os, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
W = 200
tot = K + W
L = bnr_amd.lib()
print("library:", os.environ.get("BNR_HIP_LIB", "default"), flush=True)


def digest(tab):
    h = hashlib.sha1()
    for k in sorted(tab):
        h.update(np.ascontiguousarray(tab[k]).tobytes())
    return h.hexdigest()[:12]


def group_time(nb, opts, exp=0, label=None):
    ch = bnr_amd.Chain(X, y, 7, tot, 5, 1)
    members = [ch] + [bnr_amd.Chain.like(ch, 5, c, tot) for c in range(2, nb + 1)]
    for c in members: c.init_prior()
    g = bnr_amd.Group(members) if nb > 1 else ch
    for k, v in opts.items(): g.set_option(k, v)
    g.prepare()
    g.run(2, tot, W)
    if exp:
        rc = L.bnr_debug_set_exp(0, exp)
        if rc:
            print("   (no experiments in this build)"); exp = 0
    t = time.time()
    try:
        g.run(W + 1, tot, tot)
        dt = time.time() - t
        tab = members[min(3, nb - 1)].fetch(tot, tot)
        print("%d chain(s) %-46s %7.1f us per sweep %7.0f it/s  row %s %s" % (nb, label or str(opts), 1e6 * dt / K, nb * K / dt, digest(tab) if not exp else "(timing only)",
              {k: v for k, v in ch.counters().items() if v and k != 'where'}), flush=True)
    except Exception as e:
        print("%d chain(s) %-46s FAILED: %s" % (nb, label or str(opts), e), flush=True)
    if exp: L.bnr_debug_set_exp(0, 0)
    if nb > 1: g.close()
    for c in members: c.close()


def gram_alone(nb, variant):
    tot2 = 64
    ch = bnr_amd.Chain(X, y, 7, tot2, 5, 1)
    members = [ch] + [bnr_amd.Chain.like(ch, 5, c, tot2) for c in range(2, nb + 1)]
    for c in members: c.init_prior()
    g = bnr_amd.Group(members) if nb > 1 else ch
    g.set_option("gram_variant", variant); g.set_option("overlap", 0)
    g.set_profiling(True)
    g.run(2, tot2, tot2)
    us, n = g.last_timing(1)
    print("%d chain(s) gram_variant %2d alone (single stream, eager): %.1f us over %d launches" % (nb, variant, us, n), flush=True)
    g.set_profiling(False)
    if nb > 1: g.close()
    for c in members: c.close()


for v in (8, 11, 12):
    gram_alone(8, v)
for v in (16, 8, 11):
    gram_alone(1, v)
group_time(8, {})
group_time(8, {"crit_origin": 1})
group_time(8, {"crit_origin": 2})
group_time(8, {"gram_variant": 11})
group_time(8, {"gram_variant": 12})
group_time(8, {"gram_variant": 11, "crit_origin": 1})
group_time(8, {}, exp=1, label="scalar branch returns at once")
group_time(8, {"crit_origin": 1}, exp=1, label="crit_origin 1, scalar branch returns at once")
group_time(1, {})
group_time(1, {"crit_origin": 1})
group_time(1, {}, exp=1, label="scalar branch returns at once")
