"""CPU tests that PIN the oracle (oracle/bnr_oracle.c) to the reference.

Golden vectors: tests/golden/*.npz, extracted by tests/golden/make_fixtures.py from the reference's own
test/data/gen_test_results.jld2 (+ test/data/test1.csv).  The reference's sample PATH cannot be reproduced
without Julia's Xoshiro/Distributions internals, so the pins are:
  * rhat: exact known-answer test against the golden rhat vectors (convergence.jl:4-65);
  * every full conditional: PIT/KS calibration of the golden draws under the oracle's deterministic parameters;
  * Summary known answers recorded in SURVEY.md section 4;
  * the oracle's own samplers: PIT of an oracle trace, GIG vs scipy's geninvgauss, numpy cross-check of the
    gamma update.
"""
import os

import numpy as np
import pytest
from scipy import stats

from oracle import bnr_oracle as bo
from pit import pit_trace

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P_MIN = 1e-3          # KS / chi2 p-value floor for fixed-seed calibration tests
Z_MAX = 4.0


@pytest.fixture(scope="module")
def res2():
    return dict(np.load(os.path.join(G, "golden_res2.npz")))


@pytest.fixture(scope="module")
def res():
    return dict(np.load(os.path.join(G, "golden_res.npz")))


@pytest.fixture(scope="module")
def test1():
    d = np.load(os.path.join(G, "test1_xy.npz"))
    return d["X"], d["y"]


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    assert bo.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert bo.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert bo.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


@pytest.mark.parametrize("name", ["res", "res2"])
def test_rhat_known_answer(name):
    g = dict(np.load(os.path.join(G, "golden_%s.npz" % name)))
    b, s = int(g["burn_in"]), int(g["sampled"])
    rg = bo.rhat(g["gamma"][b:b + s, :, 0][:, :, None])
    rx = bo.rhat(g["xi"][b:b + s, :, 0][:, :, None])
    assert np.allclose(rg, g["rhat_gamma"], rtol=1e-13, atol=0)
    assert np.allclose(rx, g["rhat_xi"], rtol=1e-13, atol=0)


def test_rhat_edge_cases():
    x = np.zeros((10, 2, 2))
    x[:, 1, :] = np.arange(10)[:, None]
    r = bo.rhat(x)
    assert r[0] == 1.0                     # var+ == 0 and W == 0 -> 1   (convergence.jl:55-56)
    y = np.zeros((10, 1, 2))
    y[:, 0, 1] = 1.0                       # W == 0, between-chain variance > 0 -> Inf
    assert np.isinf(bo.rhat(y)[0])
    z = np.random.default_rng(0).standard_normal((11, 3, 2))     # odd N: middle sample dropped
    r = bo.rhat(z)
    h = 5
    halves = np.concatenate([z[:h], z[-h:]], axis=2)             # (5, 3, 4)
    W = halves.var(axis=0, ddof=1).mean(axis=1)
    B = halves.mean(axis=0).var(axis=1, ddof=1)
    assert np.allclose(r, np.sqrt(((h - 1) / h * W + B) / W), rtol=1e-13)


def _check(out, skip=()):
    for k, v in out.items():
        if k.startswith("_") or k in skip:
            continue
        kind, val, n = v
        if kind == "calibration_z":
            assert abs(val) < Z_MAX, (k, v)
        else:
            assert val > P_MIN, (k, v)


@pytest.mark.parametrize("pdf_mode", [0, 1])
def test_golden_res2_conditionals_calibrated(res2, test1, pdf_mode):
    """Every conditional of SURVEY.md 8(a), evaluated by the oracle on the 399 golden transitions of res2
    (test1.csv, n=70, V=19, R=5), makes the reference's own draws uniform/normal."""
    X, y = test1
    out = pit_trace(res2, X, y, 5, pdf_mode=pdf_mode)
    _check(out)
    # the figures recorded in the survey session are reproduced
    assert abs(out["tau2"][1] - 0.86) < 0.01 and abs(out["gamma"][1] - 0.93) < 0.01 and abs(out["S"][1] - 0.35) < 0.01
    assert out["_lam_counts"][0] == [1200.0, 493.0, 302.0]
    assert out["gamma"][2] == 75810 and out["u"][2] == 16825


def test_golden_toy_conditionals_calibrated(res):
    """Toy trace (V=4, R=5): its X,y are not stored, so only the X-free conditionals are checked."""
    X = np.zeros((10, 10))
    out = pit_trace(res, X, np.zeros(10), 5, use_x=False)
    _check(out)
    assert "tau2" not in out and "gamma" not in out


def test_golden_st_rows():
    st = dict(np.load(os.path.join(G, "golden_st.npz")))
    # shapes of init-tests.jl:55-62 and the init-row conventions of gibbs.jl:199-217
    assert st["st1_u"].shape == (20, 7, 4) and st["st1_gamma"].shape == (20, 10, 1) and st["st1_pi"].shape == (20, 7, 3)
    assert st["st1_theta"][0, 0, 0] == 0.5 and st["st1_Delta"][0, 0, 0] == 0.5
    assert st["st1_mu"][0, 0, 0] == 1.0 and st["st1_tau2"][0, 0, 0] == 1.0
    assert set(np.unique(st["st1_xi"][0])) <= {0.0, 1.0}
    assert set(np.unique(st["st1_lam"][0])) <= {0.0, 1.0, -1.0}
    assert np.allclose(st["st1_pi"][0].sum(axis=1), 1.0)
    assert np.all(st["st1_S"][0] > 0)
    # the oracle's init row obeys the same conventions
    o = bo.Oracle(np.zeros((4, 10)), np.zeros(4), 7, 3, seed=100)
    o.init_prior()
    for k in ("theta", "Delta", "mu", "tau2"):
        assert o.t[k][0, 0, 0] == st["st1_" + k][0, 0, 0]
    assert np.allclose(o.t["pi"][0].sum(axis=1), 1.0) and np.all(o.t["S"][0] > 0)
    # X-free conditionals on the recorded single transitions st1 -> st2 (row 2) and st2 -> st3 (row 3)
    for name, row in (("st2", 1), ("st3", 2)):
        tbl = {k: st["%s_%s" % (name, k)] for k in bo.COLUMNS}
        out = pit_trace(tbl, np.zeros((4, 10)), np.zeros(4), 7, rows=[row], use_x=False)
        assert abs(out["xi"][1]) < Z_MAX
        assert out["u"][1] > 1e-4 and out["pi0"][1] > 1e-4


def test_summary_known_answers(res2):
    """Summary (gibbs.jl:1214-1250) known answers recorded in SURVEY.md section 4."""
    import bnr_amd
    r = bnr_amd.Results(res2, res2["rhat_xi"], res2["rhat_gamma"], int(res2["burn_in"]), int(res2["sampled"]))
    s = bnr_amd.Summary(r)
    assert np.allclose(s.edge_coef["estimate"][:5], np.round([0.121835, 1.589945, -0.007988, -1.201042, -0.606545], 3))
    assert s.edge_coef["lower_bound"][0] == round(-2.458201, 3) and s.edge_coef["upper_bound"][0] == round(3.374168, 3)
    assert np.all((s.prob_nodes["probability"] >= 0.415) & (s.prob_nodes["probability"] <= 0.50))
    assert list(s.edge_coef["node1"][:3]) == [1, 1, 1] and list(s.edge_coef["node2"][:3]) == [1, 2, 3]
    assert s.edge_coef["node1"][-1] == 19 and s.edge_coef["node2"][-1] == 19


@pytest.fixture(scope="module")
def oracle_trace(test1):
    X, y = test1
    o = bo.Oracle(X, y, 5, 600, 1234, chain=1)
    o.init_prior()
    assert o.run(2, 300, 600) == 601
    return o


def test_oracle_own_trace_calibrated(oracle_trace, test1):
    X, y = test1
    _check(pit_trace(oracle_trace.t, X, y, 5), skip=())
    assert oracle_trace.status == 0 and oracle_trace.o.nan_w_events == 0


def test_oracle_gig_branch_mix_like_golden(oracle_trace):
    """GIG branch mix of an oracle chain on test1.csv close to the golden run's (76.7 / 18.4 / 5.0 %)."""
    br = np.array(oracle_trace.o.gig_branch[:3], dtype=float)
    br /= br.sum()
    assert abs(br[1] - 0.767) < 0.1 and abs(br[2] - 0.184) < 0.1 and abs(br[0] - 0.05) < 0.05


N_REPLICATES = 200
_REPLICATE_WORKER = r"""
import os, sys
os.environ["OMP_NUM_THREADS"] = "1"
root, first, last, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from oracle import bnr_oracle as bo
from replicates import window_stats
d = np.load(os.path.join(root, "tests", "golden", "test1_xy.npz"))
rows = []
for i in range(first, last):
    o = bo.Oracle(d["X"], d["y"], 5, 400, 7000 + i, chain=1)          # reference weights (pdf_mode 0), reference defaults
    o.init_prior()
    assert o.run(2, 200, 400) == 401 and o.status == 0
    rows.append(window_stats(o.t, 200, 200))
np.save(out, np.array(rows))
"""


@pytest.fixture(scope="module")
def oracle_replicates(tmp_path_factory):
    """N_REPLICATES independent oracle runs of the golden setup (test1.csv, R=5, nburn=200, nsamp=200, one chain each, seeds
    7000+i), spread over the host cores as separate processes."""
    import subprocess
    import sys
    tmp = tmp_path_factory.mktemp("replicates")
    ncpu = max(1, min(8, len(os.sched_getaffinity(0))))
    bounds = np.linspace(0, N_REPLICATES, ncpu + 1).astype(int)
    procs = []
    for w in range(ncpu):
        procs.append(subprocess.Popen([sys.executable, "-c", _REPLICATE_WORKER, ROOT, str(bounds[w]), str(bounds[w + 1]), str(tmp / ("r%d.npy" % w))]))
    for p in procs:
        assert p.wait(timeout=900) == 0
    return np.concatenate([np.load(tmp / ("r%d.npy" % w)) for w in range(ncpu)])


def test_golden_run_is_a_typical_oracle_run(oracle_replicates, res2):
    """End-to-end pin with a stated tolerance (test/test1-generate-samples-test.jl:10-46, golden res2): every window statistic of
    the reference's stored run -- means of tau2, theta, mu, Delta, gamma[1:5], P(xi=1) (mean/min/max over nodes), max split-Rhat of
    gamma and xi, the scale of S, the share of non-zero lambda, spread and 5 %/95 % order statistics of gamma_1 -- lies inside the
    central 99 % of the distribution of that statistic over N_REPLICATES independent oracle runs of the same setup
    (rank p-value >= 0.01 each).  The replicates must also really spread (the check is not vacuous)."""
    from replicates import STAT_NAMES, assert_golden_is_typical, window_stats
    golden = window_stats(res2, 200, 200)
    assert abs(golden[0] - 0.668837) < 1e-6 and abs(golden[12] - 1.064741) < 1e-6        # SURVEY.md section 4 known answers
    assert oracle_replicates.shape == (N_REPLICATES, len(STAT_NAMES)) and np.all(np.isfinite(oracle_replicates))
    p = assert_golden_is_typical(golden, oracle_replicates, what="oracle:")
    # not vacuous: a sampler whose gamma_4 were shifted by 2, whose gamma_1 spread were 1.5 x larger, or whose tau2 level were
    # 30 x higher would be rejected by the same rule
    from replicates import rank_pvalues
    shifted = golden.copy()
    shifted[STAT_NAMES.index("mean_gamma4")] += 2.0
    shifted[STAT_NAMES.index("sd_gamma1")] *= 1.5
    shifted[STAT_NAMES.index("mean_tau2")] *= 30.0
    ps = rank_pvalues(shifted, oracle_replicates)
    for nm in ("mean_gamma4", "sd_gamma1", "mean_tau2"):
        assert ps[STAT_NAMES.index(nm)] < 0.01, (nm, ps[STAT_NAMES.index(nm)])
    assert np.median(p) > 0.1                                                              # no systematic edge-hugging


def test_pdf_modes_agree(test1):
    """Reference dense-pdf weights (gibbs.jl:349-351) == log-space weights wherever the pdfs do not underflow."""
    X, y = test1
    a = bo.Oracle(X, y, 5, 30, 7, pdf_mode=0)
    b = bo.Oracle(X, y, 5, 30, 7, pdf_mode=1)
    for o in (a, b):
        o.init_prior()
        o.run(2, 30, 30)
    for k in bo.COLUMNS:
        assert np.allclose(a.t[k], b.t[k], rtol=1e-8, atol=1e-12), k


@pytest.mark.parametrize("chi,psi", [(4.0, 9.0), (0.3, 2.0), (1e-3, 5.0), (25.0, 0.5), (1e-5, 0.3)])
def test_gig_sampler_matches_scipy(test1, chi, psi):
    """sample_gig (gig.jl) for lambda = 1/2 over the three branches (omega = sqrt(chi psi): >3 shift, (0.2,3] noshift,
    <=0.2 concave): KS against scipy's geninvgauss (density ~ x^(p-1) exp(-b(x+1/x)/2), scaled)."""
    X, y = test1
    o = bo.Oracle(X, y, 5, 4, 99)
    x = np.array([o.sample_gig(0.5, chi, psi, 3, e) for e in range(20000)])
    b, scale = np.sqrt(chi * psi), np.sqrt(chi / psi)
    p = stats.kstest(x, lambda v: stats.geninvgauss.cdf(v, 0.5, b, scale=scale)).pvalue
    assert p > P_MIN, (chi, psi, p)
    assert o.status == 0


def test_gig_degenerate_branches_keep_reference_quirk(test1):
    """chi < 10 eps -> Gamma(lambda, SCALE psi/2) (gig.jl:15-17, not GIGrvg's 2/psi); psi < 10 eps -> 1/Gamma(lambda, chi/2)."""
    X, y = test1
    o = bo.Oracle(X, y, 5, 4, 5)
    x = np.array([o.sample_gig(0.5, 1e-16, 3.0, 3, e) for e in range(8000)])
    assert stats.kstest(x, "gamma", args=(0.5, 0, 1.5)).pvalue > P_MIN
    x = np.array([o.sample_gig(0.5, 3.0, 1e-16, 3, e) for e in range(8000)])
    assert stats.kstest(1.0 / x, "gamma", args=(0.5, 0, 1.5)).pvalue > P_MIN


def test_update_gamma_matches_numpy_restatement(test1):
    """gibbs.jl:420-438 restated in numpy with the oracle's own z draws."""
    X, y = test1
    n, q = X.shape
    o = bo.Oracle(X, y, 5, 4, 321)
    o.init_prior()
    o.update("tau2", 1, 2)
    o.update("u_xi", 1, 2)
    o.update("gamma", 1, 2)
    seed = 321 + 1
    z1 = np.array([bo.normal(seed, 2, bo.SITES["G_Z1"], e) for e in range(q)])
    z2 = np.array([bo.normal(seed, 2, bo.SITES["G_Z2"], i) for i in range(n)])
    tau2 = o.t["tau2"][1, 0, 0]
    tau = np.sqrt(tau2)
    W = o.compute_W(1, 0)
    D = o.t["S"][0, :, 0]
    dg1 = np.sqrt(tau2 * D) * z1
    Xt = X / tau
    a1 = (y - X @ W - o.t["mu"][0, 0, 0]) / tau
    a3 = Xt @ dg1 + z2
    a4 = np.linalg.solve(Xt @ np.diag(tau2 * D) @ Xt.T + np.eye(n), a1 - a3)
    gam = dg1 + (tau2 * D) * (Xt.T @ a4) + W
    assert np.allclose(o.t["gamma"][1, :, 0], gam, rtol=1e-9, atol=1e-11)
    # W and the edge order (utils.jl:50-55): column-wise lower triangle including the diagonal
    u, lam = o.t["u"][1], o.t["lam"][0, :, 0]
    full = u.T @ np.diag(lam) @ u
    assert np.allclose(W, np.concatenate([full[k:, k] for k in range(19)]))


def test_reference_cost_mode_same_result(test1):
    X, y = test1
    a = bo.Oracle(X, y, 5, 6, 11, cost_mode=0)
    b = bo.Oracle(X, y, 5, 6, 11, cost_mode=1)
    for o in (a, b):
        o.init_prior()
        o.run(2, 6, 6)
    assert np.allclose(a.t["gamma"], b.t["gamma"], rtol=1e-9, atol=1e-11)


def test_purge_ring_matches_plain_run(test1):
    """run! with purge_burn (gibbs.jl:857-860): post-burn samples land in rows purge_burn+1.. and equal the plain run's
    (the draw-site contract keys variates by the global iteration, not by the ring row)."""
    X, y = test1
    nburn, nsamp, pb = 12, 6, 4
    a = bo.Oracle(X, y, 5, nburn + nsamp, 77)
    a.init_prior()
    a.run(2, nburn, nburn + nsamp)
    b = bo.Oracle(X, y, 5, nsamp + pb, 77)
    b.init_prior()
    nxt = b.run(2, nburn, nburn + nsamp, purge_burn=pb)
    assert nxt == pb + nsamp + 1
    for k in bo.COLUMNS:
        assert np.array_equal(a.t[k][nburn:nburn + nsamp], b.t[k][pb:pb + nsamp]), k


def test_continuation_equals_single_run(test1):
    X, y = test1
    a = bo.Oracle(X, y, 5, 20, 3)
    a.init_prior()
    a.run(2, 20, 20)
    b = bo.Oracle(X, y, 5, 20, 3)
    b.init_prior()
    b.run(2, 20, 9)
    b.run(10, 20, 20)
    for k in bo.COLUMNS:
        assert np.array_equal(a.t[k], b.t[k]), k
