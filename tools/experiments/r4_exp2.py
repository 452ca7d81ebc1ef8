#!/usr/bin/env python3
"""Round 4, second pass: the resident Gram with reserved CUs (k_gram8q) alone and under the pipelined schedule (factorization beside the Gram).
  BNR_HIP_LIB=<build> tools/r4_exp2.py [sweeps]"""
import sys, os, time, hashlib
os.environ.setdefault("BNR_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesiannetworkregression.jl_amd", "csrc", "_var", "exp.so"))   # the experiments build (tools/r4_build_variants.sh)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
W = 200
tot = K + W
print("library:", os.environ.get("BNR_HIP_LIB", "default"), flush=True)


def digest(tab):
    h = hashlib.sha1()
    for k in sorted(tab):
        h.update(np.ascontiguousarray(tab[k]).tobytes())
    return h.hexdigest()[:12]


def group_time(nb, opts, label=None):
    ch = bnr_amd.Chain(X, y, 7, tot, 5, 1)
    members = [ch] + [bnr_amd.Chain.like(ch, 5, c, tot) for c in range(2, nb + 1)]
    for c in members: c.init_prior()
    g = bnr_amd.Group(members) if nb > 1 else ch
    try:
        for k, v in opts.items(): g.set_option(k, v)
        g.prepare()
        g.run(2, tot, W)
        t = time.time()
        g.run(W + 1, tot, tot)
        dt = time.time() - t
        tab = members[min(3, nb - 1)].fetch(tot, tot)
        print("%d chain(s) %-70s %7.1f us per sweep %7.0f it/s  row %s %s" % (nb, label or str(opts), 1e6 * dt / K, nb * K / dt, digest(tab),
              {k: v for k, v in ch.counters().items() if v and k != 'where'}), flush=True)
    except Exception as e:
        print("%d chain(s) %-70s FAILED: %s" % (nb, label or str(opts), e), flush=True)
    try:
        if nb > 1: g.close()
        for c in members: c.close()
    except Exception as e:
        print("close failed", e)


def gram_alone(nb, variant, mask):
    tot2 = 64
    ch = bnr_amd.Chain(X, y, 7, tot2, 5, 1)
    members = [ch] + [bnr_amd.Chain.like(ch, 5, c, tot2) for c in range(2, nb + 1)]
    for c in members: c.init_prior()
    g = bnr_amd.Group(members) if nb > 1 else ch
    g.set_option("gram_variant", variant); g.set_option("overlap", 0); g.set_option("resv_mask", mask)
    g.set_profiling(True)
    g.run(2, tot2, tot2)
    us, n = g.last_timing(1)
    print("%d chain(s) gram_variant %2d resv_mask 0x%02x alone (single stream, eager): %.1f us over %d launches  %s" % (nb, variant, mask, us, n, {k: v for k, v in ch.counters().items() if v and k != 'where'}), flush=True)
    g.set_profiling(False)
    if nb > 1: g.close()
    for c in members: c.close()


gram_alone(8, 8, 0)
gram_alone(8, 11, 0)
gram_alone(8, 14, 0)
for m in (0, 0x80, 0xC0, 0xE0, 0xF0):
    gram_alone(8, 13, m)
gram_alone(1, 13, 0)
if os.environ.get("BNR_EXP2_GRAM_ONLY"): sys.exit(0)
group_time(8, {})
group_time(8, {"gram_variant": 13, "resv_mask": 0})
group_time(8, {"gram_variant": 13, "resv_mask": 0x80})
group_time(8, {"gram_variant": 13, "resv_mask": 0xC0})
group_time(8, {"factor_variant": 1})
for m in (0x80, 0xC0, 0xE0, 0xF0):
    group_time(8, {"factor_variant": 1, "pipeline": 1, "resv_mask": m})
group_time(1, {})
group_time(1, {"factor_variant": 1, "pipeline": 1, "resv_mask": 0x80})
