"""Every public entry point of the Python mirror in a process of its own (nothing loaded before it): the README example, the doubling scheme, vector-of-matrices
input with x_transform, and the refusal when torch was imported first (load order).  Run on a GPU box: python3 tools/fresh_process_api.py"""
import subprocess, sys, os, textwrap
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cases = {
 "readme_fit": """
X, y, _ = bnr_amd.make_synthetic(100, 30, 7, seed=2)
res = bnr_amd.Fit(X, y, 7, V=30, nburn=300, nsamples=200, num_chains=8, seed=1234, x_transform=False, summary_interval=95, return_state=False, ess_max_lag=0, suppress_timer=True)
print(bnr_amd.Summary(res)); print(float(res.rhatgamma.max()), float(np.nanmin(res.essgamma)))
""",
 "dbl": """
X, y, _ = bnr_amd.make_synthetic(70, 19, 5, seed=3)
res = bnr_amd.generate_samples_dbl(X, y, 5, mingen=200, maxgen=800, psrf_cutoff=1.01, num_chains=3, seed=5, x_transform=False, suppress_timer=True)
print(type(res).__name__, float(res.rhatgamma.max()))
""",
 "matrices_xtransform": """
rng = np.random.default_rng(1)
A = [ (lambda m: (m + m.T) / 2)(rng.integers(0, 2, (12, 12)).astype(float)) for _ in range(40) ]
y = rng.normal(size=40)
res = bnr_amd.Fit(A, y, 3, V=12, nburn=100, nsamples=100, num_chains=2, seed=7, suppress_timer=True, psrf_cutoff=50.0)
print(type(res).__name__, sorted(res.state.keys())[:3])
""",
 "torch_first_then_lib": """
import torch
try:
    X, y, _ = bnr_amd.make_synthetic(20, 5, 2, seed=1)
    bnr_amd.Fit(X, y, 2, V=5, nburn=10, nsamples=10, num_chains=1, seed=7, x_transform=False, suppress_timer=True)
    print("ran")
except Exception as e:
    print("refused:", str(e)[:120])
""",
}
for name, body in cases.items():
    code = "import sys, os; sys.path.insert(0, %r); import numpy as np, bnr_amd\n" % ROOT + textwrap.dedent(body)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd="/tmp")
    print("==", name, "rc", out.returncode)
    print("\n".join(out.stdout.strip().splitlines()[-6:]))
    if out.returncode: print(out.stderr[-1500:])
