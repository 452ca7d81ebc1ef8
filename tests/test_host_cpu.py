"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol of include/bnr_hip.h,
its host copies of the draw-site primitives equal the oracle's bit for bit, the split-Rhat finish equals rhat(),
the reference-interface helpers behave as the reference's, and the multi-rank Rhat exchange works over gloo.
No compute call is made (there is no GPU here); the GPU path is covered by the -m gpu tests."""
import json
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

import bnr_amd
from oracle import bnr_oracle as bo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "bnr_hip.h")).read()
    declared = set(re.findall(r"\b(bnr_[a-zA-Z0-9_]+)\s*\(", hdr)) - {"bnr_progress_cb"}
    assert len(declared) >= 35
    L = bnr_amd.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "libbnr_hip.so does not export %s" % name
    assert declared == set(bnr_amd.EXPORTS), declared ^ set(bnr_amd.EXPORTS)
    m = re.search(r"#define BNR_ABI_VERSION (\d+)", hdr)
    assert L.bnr_abi_version() == int(m.group(1)) >= 3


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product must fail loudly (status BNR_ERR_HIP), never compute on the CPU."""
    try:
        ndev = bnr_amd.device_count()
    except bnr_amd.BnrError as e:
        assert e.code == 2
        ndev = 0
    if ndev == 0:
        X, y, _ = bnr_amd.make_synthetic(8, 4, 2, seed=1)
        with pytest.raises(bnr_amd.BnrError):
            bnr_amd.Chain(X, y, 2, 4, 1, 1)
        with pytest.raises(bnr_amd.BnrError):
            bnr_amd.generate_samples(X, y, 2, nburn=2, nsamp=2, x_transform=False, num_chains=1, seed=1, suppress_timer=True)


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "bayesiannetworkregression.jl_amd")
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("not the oracle", ""), "%s mentions the oracle" % f


def test_host_rng_equals_oracle_bitwise():
    import ctypes as C
    L = bnr_amd.lib()
    rng = np.random.default_rng(5)
    out = (C.c_uint32 * 4)()
    for _ in range(50):
        c = [int(v) for v in rng.integers(0, 2**32, 4)]
        k = [int(v) for v in rng.integers(0, 2**32, 2)]
        L.bnr_host_philox((C.c_uint32 * 4)(*c), (C.c_uint32 * 2)(*k), out)
        assert list(out) == bo.philox(c, k)
    d = np.load(os.path.join(G, "test1_xy.npz"))
    o = bo.Oracle(d["X"], d["y"], 5, 4, 1000, chain=1)
    seed = 1001
    u2 = (C.c_double * 2)()
    for e in range(300):
        L.bnr_host_uniform2(seed, 7, 21, e, 3, u2)
        assert (u2[0], u2[1]) == bo.uniform2(seed, 7, 21, e, 3)
        assert 0.0 < u2[0] < 1.0 and 0.0 < u2[1] < 1.0
        assert L.bnr_host_normal(seed, 5, 19, e, 0) == bo.normal(seed, 5, 19, e, 0)
        for sh in (0.5, 1.0, 2.5, 130.5):
            assert L.bnr_host_gamma(seed, sh, 5, 23, e) == o.gamma_draw(sh, 5, 23, e)
    for e in range(2000):
        chi, psi = 10 ** rng.uniform(-7, 2), 10 ** rng.uniform(-2, 1)
        a, b = L.bnr_host_gig(seed, 0.5, chi, psi, 5, e), o.sample_gig(0.5, chi, psi, 5, e)
        # same variates, same accept/reject decisions; the product evaluates x^(1/2), x^(-1/2) as square roots where the
        # oracle calls pow() like the reference (gig.jl:108-141): last-bit differences only
        assert abs(a - b) <= 1e-12 * abs(b)
    assert sum(o.o.gig_branch[:3]) == 2000 and min(o.o.gig_branch[:3]) > 50      # all three branches exercised
    for V in (2, 5, 19):
        e = 0
        for k in range(V):
            for l in range(k, V):
                assert L.bnr_host_edge_index(V, l, k) == e == L.bnr_host_edge_index(V, k, l)
                e += 1


def _stats_from_table(g, xi, first, nsamp):
    """numpy restatement of bnr_chain_rhat_stats' message for one chain."""
    par = np.concatenate([g[first:first + nsamp, :, 0], xi[first:first + nsamp, :, 0]], axis=1)
    h = nsamp // 2
    a, b = par[:h], par[nsamp - h:]
    return np.concatenate([a.mean(0), a.var(0, ddof=1), b.mean(0), b.var(0, ddof=1)])


@pytest.mark.parametrize("name", ["res", "res2"])
def test_rhat_from_stats_known_answer(name):
    g = dict(np.load(os.path.join(G, "golden_%s.npz" % name)))
    b, s = int(g["burn_in"]), int(g["sampled"])
    st = _stats_from_table(g["gamma"], g["xi"], b, s)[None, :]
    r = bnr_amd.rhat_from_stats(st, s)
    q = g["gamma"].shape[1]
    assert np.allclose(r[:q], g["rhat_gamma"], rtol=1e-12) and np.allclose(r[q:], g["rhat_xi"], rtol=1e-12)


def test_rhat_from_stats_multichain_equals_oracle_rhat():
    rng = np.random.default_rng(2)
    nsamp, q, V, C = 41, 6, 3, 3
    gam = rng.standard_normal((C, nsamp, q, 1)) + np.arange(C)[:, None, None, None] * 0.3
    xi = (rng.random((C, nsamp, V, 1)) < 0.4).astype(float)
    xi[:, :, 0, :] = 1.0                                     # constant parameter -> Rhat = 1
    st = np.stack([_stats_from_table(gam[c], xi[c], 0, nsamp) for c in range(C)])
    r = bnr_amd.rhat_from_stats(st, nsamp)
    ref_g = bo.rhat(np.transpose(gam[:, :, :, 0], (1, 2, 0)))
    ref_x = bo.rhat(np.transpose(xi[:, :, :, 0], (1, 2, 0)))
    assert np.allclose(r[:q], ref_g, rtol=1e-12) and np.allclose(r[q:], ref_x, rtol=1e-12)
    assert r[q] == 1.0


def _ess_message(x, L):
    """numpy restatement of bnr_chain_ess_stats' message for one chain; x: (nsamp, nparams)."""
    n = x.shape[0]
    h = n // 2
    out = []
    for half in (x[:h], x[n - h:]):
        m = half.mean(0)
        c = half - m
        ac = np.stack([(c[:h - t] * c[t:]).sum(0) / h for t in range(L)])
        out.append(np.concatenate([m[None], (c * c).sum(0)[None] / (h - 1), ac]))
    return np.concatenate(out).ravel()


def _ess_reference(chains, L):
    """Independent numpy restatement of the split-chain Geyer estimator (Stan's compute_effective_sample_size), one parameter."""
    halves = []
    for x in chains:
        n = len(x); h = n // 2
        halves += [x[:h], x[n - h:]]
    h = len(halves[0]); m = len(halves)
    means = np.array([v.mean() for v in halves]); vars_ = np.array([v.var(ddof=1) for v in halves])
    W = vars_.mean(); varp = W * (h - 1) / h + means.var(ddof=1)
    acov = np.array([[((v - v.mean())[:h - t] * (v - v.mean())[t:]).sum() / h for t in range(L)] for v in halves]).mean(0)
    rho = 1 - (W - acov) / varp
    rho[0] = 1.0
    tau, prev, t = -1.0, np.inf, 0
    while 2 * t + 1 < L:
        P = rho[2 * t] + rho[2 * t + 1]
        if not P > 0:
            break
        P = min(P, prev); prev = P; tau += 2 * P; t += 1
    if 2 * t < L and rho[2 * t] > 0:
        tau += rho[2 * t]
    tau = max(tau, 1 / np.log10(m * h))
    return m * h / tau


def test_ess_from_stats_matches_restatement_and_ar1_theory():
    """bnr_ess_from_stats (an addition to the reference, which only has split-Rhat): equals an independent numpy
    restatement of the split-chain Geyer estimator and recovers the effective sample size of AR(1) chains,
    N (1 - phi) / (1 + phi)."""
    rng = np.random.default_rng(7)
    nch, n, L = 4, 4000, 200
    phis = [0.0, 0.5, 0.9, -0.3]
    x = np.zeros((nch, n, len(phis) + 1))
    for c in range(nch):
        for j, phi in enumerate(phis):
            e = rng.standard_normal(n) * np.sqrt(1 - phi * phi)
            v = np.empty(n); v[0] = rng.standard_normal()
            for i in range(1, n):
                v[i] = phi * v[i - 1] + e[i]
            x[c, :, j] = v
        x[c, :, -1] = 3.0                                        # constant parameter -> NaN
    st = np.stack([_ess_message(x[c], L) for c in range(nch)])
    ess = bnr_amd.ess_from_stats(st, n, L)
    assert np.isnan(ess[-1])
    for j, phi in enumerate(phis):
        ref = _ess_reference([x[c, :, j] for c in range(nch)], L)
        assert abs(ess[j] - ref) <= 1e-9 * ref, (phi, ess[j], ref)
        theory = nch * n * (1 - phi) / (1 + phi)
        assert 0.75 * theory < ess[j] < 1.3 * theory, (phi, ess[j], theory)


def test_lower_triangle_and_setup_X():
    A = np.arange(16.0).reshape(4, 4)
    v = bnr_amd.lower_triangle(A)                            # reads matrix[j,i], j >= i  (utils.jl:50-55)
    assert list(v) == [0, 4, 8, 12, 5, 9, 13, 10, 14, 15]
    B = bnr_amd.create_lower_tri(v, 4)
    assert np.array_equal(B, np.tril(A))
    mats = [np.tril(np.random.default_rng(i).random((5, 5))) for i in range(3)]
    Xn, V, q = bnr_amd.setup_X(mats, True)
    assert (V, q) == (5, 15) and Xn.shape == (3, 15) and np.array_equal(Xn[1], bnr_amd.lower_triangle(mats[1]))
    Xb = (np.random.default_rng(0).random((6, 10)) < 0.5)   # Bool input is converted to Float64 (toy test uses Bool X)
    Xn, V, q = bnr_amd.setup_X(Xb, False)
    assert (V, q) == (4, 10) and Xn.dtype == np.float64
    ex = np.load(os.path.join(G, "examples_xy.npz"))
    Xn, V, q = bnr_amd.setup_X(ex["X"], False)
    assert (V, q) == (30, 465)


def test_synthetic_generator_is_deterministic():
    X1, y1, t1 = bnr_amd.make_synthetic(20, 6, 3, seed=3)
    X2, y2, t2 = bnr_amd.make_synthetic(20, 6, 3, seed=3)
    assert np.array_equal(X1, X2) and np.array_equal(y1, y2) and X1.shape == (20, 21)
    assert 0.3 < (X1 == 0).mean() < 0.7 and X1.min() >= 0


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import bnr_amd
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[3]
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")        # one node: do not depend on what the hostname resolves to
dist.init_process_group("gloo", rank=rank, world_size=world)
# the ONE seed of a fit: drawn on rank 0, identical on every rank (gibbs.jl:739, 928); an explicit seed passes through
import random
s1 = bnr_amd.shared_seed(None, lambda: random.SystemRandom().randrange(1, 2**31))
t = torch.tensor([s1], dtype=torch.int64); lst = [torch.zeros_like(t) for _ in range(world)]; dist.all_gather(lst, t)
assert all(int(v.item()) == s1 for v in lst), lst
assert bnr_amd.shared_seed(1234, lambda: 1 / 0) == 1234
# the library's communicator with the host's transport behind it (callback seam of bnr_comm_*): gloo here, RCCL on GPUs
comm = bnr_amd.make_comm()
got = comm.allgather(np.arange(5.0) + 10 * rank)
assert got.shape == (world, 5) and all(np.array_equal(got[r], np.arange(5.0) + 10 * r) for r in range(world)), got
comm.close()
num_chains, width = int(sys.argv[4]), 4 * 9
ids = bnr_amd.local_chain_ids(num_chains)
assert ids == [c for c in range(1, num_chains + 1) if (c - 1) % world == rank], ids
rng = lambda c: np.random.default_rng(100 + c).random(width)
local = {{c: rng(c) for c in ids}}
allst = bnr_amd.allgather_stats(local, num_chains)
ref = np.stack([rng(c) for c in range(1, num_chains + 1)])
assert np.array_equal(allst, ref)
r = bnr_amd.rhat_from_stats(allst, 20)
np.save(sys.argv[5] + ".%d.npy" % rank, r)
dist.barrier(); dist.destroy_process_group()
"""


@pytest.mark.parametrize("num_chains,world", [(2, 2), (3, 2), (1, 2), (8, 3)])
def test_rhat_exchange_two_ranks_gloo(tmp_path, num_chains, world):
    """World size 2 over gloo: chains sharded round-robin, per-chain messages all-gathered, every rank finishes the
    same Rhat (the RCCL path of bench.py / generate_samples uses the same code with backend nccl).  (8, 3): the fit's 8 chains on
    three ranks -- an uneven 3 / 3 / 2 split (gibbs.jl:946-957, 771-789)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    out = str(tmp_path / "rhat")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(port), str(num_chains), out]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    r0 = np.load(out + ".0.npy")
    for r in range(1, world):
        assert np.array_equal(r0, np.load(out + ".%d.npy" % r))
    if (num_chains, world) == (8, 3):
        assert [len([c for c in range(1, 9) if (c - 1) % 3 == r]) for r in range(3)] == [3, 3, 2]
    ref = np.stack([np.random.default_rng(100 + c).random(36) for c in range(1, num_chains + 1)])
    assert np.allclose(r0, bnr_amd.rhat_from_stats(ref, 20))


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus N` as the driver types it (no launcher, no WORLD_SIZE): the script itself starts one fresh process per GPU --
    before anything touched the GPU -- with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, chain c on rank (c-1) % N (the reference's pmap
    spawns its workers itself, gibbs.jl:946-948), and relays rank 0's ONE line.  --dry-ranks: no GPU, every rank only reports what it
    would hold (the ranks still meet over gloo on the loopback)."""
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-ranks"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and len(d["dry_ranks"]) == 2
    r0, r1 = sorted(d["dry_ranks"], key=lambda r: r["rank"])
    assert (r0["rank"], r0["local_rank"], r0["world"], r0["chain_ids"]) == (0, 0, 2, [1, 3, 5, 7])
    assert (r1["rank"], r1["local_rank"], r1["world"], r1["chain_ids"]) == (1, 1, 2, [2, 4, 6, 8])
    assert r0["master"] == r1["master"] and r0["master"].startswith("127.0.0.1:")
    # three ranks, five chains: the round-robin of api.local_chain_ids
    out = subprocess.run([sys.executable, bench, "--gpus", "3", "--chains", "5", "--dry-ranks"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip())
    assert [r["chain_ids"] for r in sorted(d["dry_ranks"], key=lambda r: r["rank"])] == [[1, 4], [2, 5], [3]]
    assert d["chains_held"] == [2, 2, 1] and d["rccl_ranks"] == 0
    # a launcher that started a different number of ranks than --gpus says: refused, loudly
    out = subprocess.run([sys.executable, bench, "--gpus", "4", "--dry-ranks"], env=dict(env, WORLD_SIZE="2", RANK="0"), capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "must agree" in out.stderr


def test_bench_eight_ranks_dry_and_a_rank_that_dies_at_start_up():
    """The N = 8 layout of BASELINE configs[2] (one chain per GPU) through the same launcher, rendezvous and gather as the real run, without a
    GPU: eight ranks, chain c on rank c - 1, the line carries what every rank holds.  And the launcher watches ALL its children: a rank that
    dies before the rendezvous ends the run at once with that rank's exit code (not after torch's rendezvous timeout)."""
    import time
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, bench, "--gpus", "8", "--dry-ranks"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip())
    assert d["n_gpus"] == 8 and d["chains_held"] == [1] * 8 and d["rccl_ranks"] == 0
    assert [r["chain_ids"] for r in sorted(d["dry_ranks"], key=lambda r: r["rank"])] == [[c] for c in range(1, 9)]
    assert sorted(r["local_rank"] for r in d["dry_ranks"]) == list(range(8))
    t0 = time.time()
    out = subprocess.run([sys.executable, bench, "--gpus", "3", "--dry-ranks"], env=dict(env, BNR_BENCH_DRY_FAIL_RANK="2"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 7 and "rank 2 exited with code 7" in out.stderr, (out.returncode, out.stderr[-2000:])
    assert time.time() - t0 < 120


def test_chains_are_refused_when_a_foreign_hip_runtime_was_loaded_first(tmp_path):
    """Load order: a process serves every libamdhip64 user from the first copy mapped.  With torch (which carries its own
    ROCm runtime) imported BEFORE the library, chain creation is refused with a message that says so; the GPU-free entry points
    stay usable; with the library loaded first everything is as usual."""
    script = tmp_path / "order.py"
    script.write_text(r'''
import sys
sys.path.insert(0, %r)
import numpy as np
first = sys.argv[1]
if first == "torch":
    import torch
import bnr_amd
from bnr_amd import _capi
L = _capi.lib()
assert L.bnr_abi_version() >= 1
if first != "torch":
    import torch
print("FOREIGN" if _capi._foreign_hip else "OWN")
assert _capi.rhat_from_stats(np.random.default_rng(1).random((2, 8)), 20).shape == (2,)       # GPU-free: always available
X, y, _ = bnr_amd.make_synthetic(8, 3, 2, seed=1)
try:
    bnr_amd.Chain(X, y, 2, 4, 1, 1)
    print("CREATED")
except _capi.BnrError as e:
    print("REFUSED" if "BEFORE importing torch" in str(e) else "OTHER: " + str(e))
''' % ROOT)
    out = subprocess.run([sys.executable, str(script), "torch"], capture_output=True, text=True, timeout=300).stdout.split()
    has_foreign = out[0] == "FOREIGN"               # only if the torch wheel really ships a runtime of its own
    if has_foreign:
        assert out[1] == "REFUSED", out
    out = subprocess.run([sys.executable, str(script), "lib"], capture_output=True, text=True, timeout=300).stdout.split()
    assert out[0] == "OWN" and out[1] in ("CREATED", "OTHER:"), out      # OTHER: no GPU in this container (BNR_ERR_HIP from hipGetDeviceCount)


def test_fit_in_a_fresh_process_does_not_pull_in_torch():
    """The README's first example -- `import bnr_amd; bnr_amd.Fit(...)` in a fresh process -- must reach the library with nothing else
    loaded: the chain-placement helpers look for a process group the CALLER initialised and never import torch themselves (importing
    it would map the torch wheel's own HIP runtime in front of the library's and trip the load-order guard above)."""
    code = ("import sys; sys.path.insert(0, %r); import bnr_amd\n"
            "X, y, _ = bnr_amd.make_synthetic(20, 5, 2, seed=1)\n"
            "try:\n"
            "    bnr_amd.generate_samples(X, y, 2, nburn=4, nsamp=4, num_chains=2, seed=3, x_transform=False, suppress_timer=True)\n"
            "    print('RAN')\n"
            "except Exception as e:\n"
            "    print('RAISED', str(e))\n"
            "print('TORCH' if 'torch' in sys.modules else 'NOTORCH')\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300).stdout
    assert "NOTORCH" in out, out
    assert "another HIP runtime" not in out, out            # (without a GPU the call ends in BNR_ERR_HIP from hipGetDeviceCount: no CPU fallback)
    assert out.startswith("RAN") or "no ROCm-capable device" in out or "HIP" in out, out


def test_a_capped_draw_does_not_abort_a_fit():
    """the C ABI reports BNR_ERR_SAMPLER_CAP (status 4) by the call it happened in, but the rows are written and the table stays valid
    (the reference's rejection loops are unbounded and never raise).  The fit drivers (generate_samples, generate_samples_dbl, Fit -- all through
    ChainSet.run) therefore warn and go on instead of dying of one capped draw in tens of thousands of iterations; any other status still raises.
    Host logic only: the chains are stand-ins that raise what the library would."""
    from bnr_amd import api, _capi

    class Fake:
        def __init__(self, code): self.code, self.calls = code, 0
        def run(self, *a):
            self.calls += 1
            if self.code: raise _capi.BnrError(self.code, "libbnr_hip: stand-in")

    for group in (False, True):
        cs = api.ChainSet.__new__(api.ChainSet)
        capped, fine = Fake(_capi.BNR_ERR_SAMPLER_CAP), Fake(0)
        cs.ids, cs.chains, cs.group = [1, 2], {1: capped, 2: fine}, (capped if group else None)
        with pytest.warns(RuntimeWarning, match="attempt cap"):
            cs.run(2, 10, 10, 0)
        assert capped.calls == 1 and (group or fine.calls == 1)            # the other chains of the rank still run
        broken = Fake(_capi.BNR_ERR_CHOLESKY)
        cs.chains, cs.group = {1: broken, 2: fine}, (broken if group else None)
        with pytest.raises(_capi.BnrError):
            cs.run(2, 10, 10, 0)



def test_late_kernels_sit_behind_the_sweep_kernels_in_the_code_object(tmp_path):
    """Round 5, notes Y: a chain alone runs 1.5 % slower when kernels added later sit in the middle of the code object (every kernel of the sweep 1.4-2.9 % slower at
    unchanged source).  The instantiations are emitted in the order of their first reference in bnr_hip.hip, so every reference to k_xpass_group2, k_backproj64 and
    k_tail<.., false> lives at the end of that file: in the gfx950 code object of the built library they must come behind every kernel of the sweep."""
    llvm = "/opt/rocm/lib/llvm/bin"
    lib = os.path.join(ROOT, "bayesiannetworkregression.jl_amd", "libbnr_hip.so")
    if not (os.path.exists(os.path.join(llvm, "clang-offload-bundler")) and os.path.exists(lib)):
        pytest.skip("no ROCm LLVM tools / no built library here")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "co.o")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"], check=True)
    out = subprocess.run([os.path.join(llvm, "llvm-readelf"), "-sW", co], check=True, stdout=subprocess.PIPE, text=True).stdout
    names = subprocess.run(["c++filt"], input=out, check=True, stdout=subprocess.PIPE, text=True).stdout
    addr = {}
    for line in names.splitlines():
        m = re.match(r"\s*\d+:\s+([0-9a-f]+)\s+\d+\s+FUNC\s+\S+\s+\S+\s+\S+\s+(?:void )?(k_\w+(?:<[^>]*>)?)", line)
        if m:
            addr[m.group(2)] = int(m.group(1), 16)
    late = {k: v for k, v in addr.items() if k.startswith(("k_xpass_group2", "k_backproj64")) or (k.startswith("k_tail<") and "false" in k)}
    sweep = {k: v for k, v in addr.items() if k not in late and k.startswith(("k_chol_step", "k_gram", "k_solve", "k_rhs", "k_xpass", "k_backproj", "k_node", "k_tail", "k_sdigits"))}
    assert len(late) >= 5 and len(sweep) >= 20, (sorted(late), len(sweep))
    assert min(late.values()) > max(sweep.values()), (sorted(late.items(), key=lambda kv: kv[1])[:2], sorted(sweep.items(), key=lambda kv: -kv[1])[:2])


def test_gram_k_split_keeps_every_k_group_inside_the_buffer_window():
    """the Gram loops address X through a 2 GiB buffer resource with 32-bit scalar offsets (csrc/bnr_kernels.h, bnr_gram16_task /
    bnr_gram8_task; reference product gibbs.jl:434).  The K split must be raised until one K-group's slice + prefetch distance fits that window --
    n = 14 000 with V >= 280 and n = 8 000 with V >= 520 used to overflow silently.  Pure host arithmetic: no GPU."""
    import ctypes as C
    L = bnr_amd.lib()
    out = (C.c_int32 * 4)()
    # the shapes every earlier round ran: the split that fills the chip is kept as it was
    for (n, V, want) in [(500, 100, 7), (200, 50, None), (2000, 200, None), (500, 300, None)]:
        assert L.bnr_host_gram_plan(n, V, 256, out) == 0
        ks, kchunk, q_pad, mib = out[0], out[1], out[2], out[3]
        q = V * (V + 1) // 2
        assert q_pad == ks * kchunk >= q and kchunk % 16 == 0
        if want is not None:
            assert ks == want
        assert mib < 2048
    # at and beyond the old silent-overflow boundary
    for (n, V) in [(14000, 280), (14000, 400), (8000, 520), (8000, 700), (14000, 1000)]:
        assert L.bnr_host_gram_plan(n, V, 256, out) == 0
        ks, kchunk, q_pad = out[0], out[1], out[2]
        n_pad = (n + 63) // 64 * 64
        span = (kchunk // 2 + 64 + 16) * n_pad * 8         # bnr_gram_span_bytes: the K-group's columns, the prefetch distance, the widest lane offset
        assert span <= 0x7FFFFFFF, (n, V, ks, span)
        assert q_pad >= V * (V + 1) // 2
        # and not split further than needed: one slice fewer would not fit (or the chip-filling split was already fine)
        if ks > 32:
            k1 = -(-(V * (V + 1) // 2) // (ks - 1))
            k1 = (k1 + 15) // 16 * 16
            assert (k1 // 2 + 80) * n_pad * 8 > 0x7FFFFFFF
    assert L.bnr_host_gram_plan(0, 5, 256, out) != 0
