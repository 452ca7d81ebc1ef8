"""One chain at the headline shape for a kernel trace (tools/prof_cmd.sh one tools/one_chain_run.py): 400 graph-replayed sweeps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bnr_amd
X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
ch = bnr_amd.Chain(X, y, 7, 420, 5, 1)
ch.init_prior()
for kv in sys.argv[1:]:
    k, v = kv.split("="); ch.set_option(k, int(v))
ch.prepare()
ch.run(2, 420, 420)
print(ch.counters())
