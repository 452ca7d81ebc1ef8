"""Experiment: G lockstep groups of C chains each driven from G host threads (do their latency-bound and
throughput-bound phases overlap on the GPU?).  usage: two_groups.py G C [steps]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bnr_amd
G, C = int(sys.argv[1]), int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 400
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
tot = K + 60
first = bnr_amd.Chain(X, y, 7, tot, 20240501, 1)
groups, allch = [], [first]
for g in range(G):
    chs = []
    for c in range(C):
        cid = g * C + c + 1
        ch = first if cid == 1 else bnr_amd.Chain.like(first, 20240501, cid, tot)
        if cid != 1: allch.append(ch)
        ch.init_prior(); chs.append(ch)
    groups.append(bnr_amd.Group(chs) if C > 1 else chs[0])
OPTS = dict(kv.split("=") for kv in sys.argv[4:])
for g in groups:
    for k, v in OPTS.items():
        g.set_option(k, int(v))
def run(g, a, b):
    try:
        g.run(a, b, b)
    except Exception as e:
        print("thread error:", e)
def run_all(a, b):
    th = [threading.Thread(target=run, args=(g, a, b)) for g in groups]
    for t in th: t.start()
    for t in th: t.join()
run_all(2, 49)
t0 = time.perf_counter()
run_all(50, 49 + K)
dt = time.perf_counter() - t0
print("%d group(s) x %d chains: %.0f it/s (%.1f us per sweep of all %d chains)" % (G, C, G * C * K / dt, 1e6 * dt / K, G * C))
print(allch[0].counters())
