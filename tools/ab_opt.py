#!/usr/bin/env python3
"""us per sweep of a chain alone or a lockstep group under chain options: tools/ab_opt.py <chains> <sweeps> [n V R] -- name=value[,name=value...] ...
(every option set after `--` is timed twice, in the order given; config default: n=500 V=100 R=5)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bnr_amd

def main():
    argv = sys.argv[1:]
    cut = argv.index("--")
    head, sets = argv[:cut], argv[cut + 1:]
    C, K = int(head[0]), int(head[1])
    n, V, R = (int(head[2]), int(head[3]), int(head[4])) if len(head) >= 5 else (500, 100, 5)
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=20240501)
    tot = 2 * K + 70
    for rep in range(2):
        for sset in sets:
            chains = []
            for c in range(1, C + 1):
                ch = bnr_amd.Chain(X, y, R, tot, 3, c) if not chains else bnr_amd.Chain.like(chains[0], 3, c, tot)
                ch.init_prior()
                chains.append(ch)
            runner = bnr_amd.Group(chains) if C > 1 else chains[0]
            if sset != "default":
                for kv in sset.split(","):
                    k, v = kv.split("=")
                    runner.set_option(k, int(v))
            runner.prepare(); runner.prepare()
            runner.run(2, 65, 65)
            bnr_amd.device_synchronize(0)
            t0 = time.perf_counter()
            runner.run(66, 65 + K, 65 + K)
            bnr_amd.device_synchronize(0)
            dt = time.perf_counter() - t0
            print("%-40s %2d chain(s) %8.1f us per sweep" % (sset, C, 1e6 * dt / K), flush=True)
            if C > 1:
                runner.close()
            for ch in chains:
                ch.close()

if __name__ == "__main__":
    main()
