"""GPU parity tests (run with -m gpu on an MI355X): the HIP path through the C ABI against the CPU oracle on the same
seeded inputs, against the golden fixtures, and -- at BASELINE.json's full sizes -- through size-independent
properties.

Tolerance (floating point, stated by north_star as an MCMC tolerance): the draw-site contract makes the HIP sampler
consume the very same variates as the oracle, so the bar used here is far tighter than distributional agreement:
every state column of every row within RTOL = 1e-6 relative (+ATOL) of the oracle over whole short runs (observed:
<= 2e-9), discrete columns (xi, lambda) exactly equal.  Distributional agreement with the REFERENCE is then inherited from
the oracle's own pins (tests/test_oracle_golden.py) and re-checked directly by PIT calibration of a GPU trace.
"""
import os

import numpy as np
import pytest

import bnr_amd
from oracle import bnr_oracle as bo
from pit import pit_trace

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL, ATOL = 1e-6, 1e-9
HOOKS = ["tau2", "u_xi", "gamma", "D", "theta", "Delta", "M", "mu", "Lambda", "pi"]
HOOK_COLS = dict(tau2=["tau2"], u_xi=["u", "xi"], gamma=["gamma"], D=["S"], theta=["theta"], Delta=["Delta"], M=["M"],
                 mu=["mu"], Lambda=["lam"], pi=["pi"])


def assert_tables_close(got, ref, rows_got=slice(None), rows_ref=slice(None), what=""):
    for k in bo.COLUMNS:
        a, b = got[k][rows_got], ref[k][rows_ref]
        assert a.shape == b.shape, (what, k, a.shape, b.shape)
        assert np.all(np.isfinite(a)), (what, k, "non-finite")
        if k in ("xi", "lam"):
            assert np.array_equal(a, b), (what, k, "discrete column differs")
        else:
            err = np.abs(a - b) / (ATOL / RTOL + np.abs(b))
            assert err.max() < RTOL, (what, k, float(err.max()))


def pair(X, y, R, tot, seed, chain=1, **hyper):
    ch = bnr_amd.Chain(X, y, R, tot, seed, chain, device=0, **hyper)
    o = bo.Oracle(X, y, R, tot, seed, chain=chain, pdf_mode=1, **hyper)
    return ch, o


@pytest.fixture(scope="module")
def test1():
    d = np.load(os.path.join(G, "test1_xy.npz"))
    return d["X"], d["y"]


CASES = [  # n, V, R, rows, normal_x   (ragged n vs the 64-row tiles, tiny V, R = 1, R > V ...)
    (40, 8, 3, 12, False), (70, 19, 5, 30, True), (1, 2, 1, 6, False), (3, 2, 4, 6, True), (65, 5, 2, 8, False),
    (129, 12, 7, 8, True), (200, 50, 5, 10, False), (64, 33, 10, 6, True),
    (1000, 10, 3, 5, True),      # 32 factorization panels; the trailing update runs in its 64 x 64 super-block form
]


@pytest.mark.parametrize("n,V,R,tot,normal_x", CASES)
def test_full_sweep_matches_oracle(gpu, n, V, R, tot, normal_x):
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=1000 + n + V, normal_x=normal_x)
    ch, o = pair(X, y, R, tot, 4242)
    ch.init_prior()
    o.init_prior()
    assert_tables_close(ch.fetch(1, 1), o.t, rows_ref=slice(0, 1), what="init row")
    nxt = ch.run(2, tot, tot)
    assert nxt == tot + 1 == o.run(2, tot, tot)
    assert_tables_close(ch.fetch(), o.t, what="run n=%d V=%d R=%d" % (n, V, R))
    c = ch.counters()
    assert c["chol_fail"] == 0 and c["sampler_cap"] == 0 and c["nan_w"] == 0
    ch.close()


@pytest.mark.parametrize("n,V,R,normal_x", [(40, 8, 3, False), (70, 19, 5, True), (4, 4, 7, True)])
def test_deconstructed_sweep_matches_oracle(gpu, n, V, R, normal_x):
    """Each update_*! called separately in sweep order on one row, as test/init-tests.jl:96-124 does."""
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=31, normal_x=normal_x)
    ch, o = pair(X, y, R, 5, 123)
    ch.init_prior()
    o.init_prior()
    for row in (2, 3):
        for name in HOOKS:
            ch.update(name, row, row)
            o.update(name, row - 1, row)
            g = ch.fetch(row, row)
            for k in HOOK_COLS[name]:
                a, b = g[k][0], o.t[k][row - 1]
                assert np.allclose(a, b, rtol=RTOL, atol=ATOL), (row, name, k)
    # one gibbs_sample! on the next row (init-tests.jl:76) continues from the hook-built rows
    ch.gibbs_step(4, 4)
    o.gibbs_sample(3, 4)
    assert_tables_close(ch.fetch(4, 4), o.t, rows_ref=slice(3, 4), what="gibbs_step")
    ch.close()


def test_test1_csv_trace_matches_oracle_and_is_calibrated(gpu, test1):
    """test/data/test1.csv (n=70, V=19, R=5; the only reference suite that can fail CI): 400 rows on the GPU equal the
    oracle's, and the GPU trace passes the same PIT calibration the golden reference trace passes."""
    X, y = test1
    ch, o = pair(X, y, 5, 400, 1234)
    ch.init_prior()
    o.init_prior()
    ch.run(2, 200, 400)
    o.run(2, 200, 400)
    got = ch.fetch()
    assert_tables_close(got, o.t, what="test1")
    out = pit_trace(got, X, y, 5)
    for k, v in out.items():
        if k.startswith("_"):
            continue
        if v[0] == "calibration_z":
            assert abs(v[1]) < 4.0, (k, v)
        else:
            assert v[1] > 1e-3, (k, v)
    # split-Rhat on the device == rhat() of the oracle on the fetched table (convergence.jl:4-65)
    st = ch.rhat_stats(201, 200)
    r = bnr_amd.rhat_from_stats(st[None, :], 200)
    assert np.allclose(r[:190], bo.rhat(got["gamma"][200:400, :, 0][:, :, None]), rtol=1e-10)
    assert np.allclose(r[190:], bo.rhat(got["xi"][200:400, :, 0][:, :, None]), rtol=1e-10)
    ch.close()


def test_golden_run_is_a_typical_gpu_run(gpu, test1):
    """The end-to-end pin of tests/test_oracle_golden.py::test_golden_run_is_a_typical_oracle_run repeated with GPU chains: 200
    independent HIP runs of the golden setup (test/test1-generate-samples-test.jl:10-12: test1.csv, R=5, nburn=200, nsamp=200, one
    chain per run, seeds 7000+i; here all advanced as ONE lockstep group) -- every window statistic of the reference's stored run
    `res2` lies inside the central 99 % of the GPU replicate distribution (rank p >= 0.01).  Three of the runs are also compared
    with the oracle run of the same seed (reference dense-pdf weights there, log-space weights here)."""
    from replicates import STAT_NAMES, assert_golden_is_typical, window_stats
    X, y = test1
    N = 200
    chains = [bnr_amd.Chain(X, y, 5, 400, 7000, 1)]
    chains += [bnr_amd.Chain.like(chains[0], 7000 + i, 1, 400) for i in range(1, N)]
    for c in chains:
        c.init_prior()
    grp = bnr_amd.Group(chains)
    assert grp.run(2, 200, 400) == 401
    reps = np.array([window_stats(c.fetch(), 200, 200) for c in chains])
    g = np.load(os.path.join(G, "golden_res2.npz"))
    golden = window_stats({k: g[k] for k in bo.COLUMNS}, 200, 200)
    assert reps.shape == (N, len(STAT_NAMES)) and np.all(np.isfinite(reps))
    p = assert_golden_is_typical(golden, reps, what="GPU:")
    assert np.median(p) > 0.1
    for i in (0, 57, 199):
        o = bo.Oracle(X, y, 5, 400, 7000 + i, chain=1, pdf_mode=0)
        o.init_prior()
        o.run(2, 200, 400)
        assert_tables_close(chains[i].fetch(), o.t, what="replicate %d vs oracle" % i)
    assert all(c.counters()["chol_fail"] == 0 for c in chains[:5])
    grp.close()
    for c in chains:
        c.close()


def test_purge_ring_and_continuation(gpu, test1):
    X, y = test1
    nburn, nsamp, pb = 12, 6, 4
    plain = bnr_amd.Chain(X, y, 5, nburn + nsamp, 77, 1)
    plain.init_prior()
    plain.run(2, nburn, nburn + nsamp)
    ring = bnr_amd.Chain(X, y, 5, nsamp + pb, 77, 1)
    ring.init_prior()
    assert ring.run(2, nburn, nburn + nsamp, purge_burn=pb) == pb + nsamp + 1
    o = bo.Oracle(X, y, 5, nsamp + pb, 77, pdf_mode=1)
    o.init_prior()
    o.run(2, nburn, nburn + nsamp, purge_burn=pb)
    a, b = plain.fetch(), ring.fetch()
    assert_tables_close(b, o.t, what="ring vs oracle ring")                  # incl. row 1 = last wrapped row
    for k in bo.COLUMNS:
        assert np.array_equal(a[k][nburn:nburn + nsamp], b[k][pb:pb + nsamp]), k
    # a run split over two calls (first_index > 2) equals the single call bit for bit
    two = bnr_amd.Chain(X, y, 5, nburn + nsamp, 77, 1)
    two.init_prior()
    assert two.run(2, nburn, 9) == 10
    assert two.run(10, nburn, nburn + nsamp) == nburn + nsamp + 1
    c = two.fetch()
    for k in bo.COLUMNS:
        assert np.allclose(a[k], c[k], rtol=1e-9, atol=1e-12), k
    with pytest.raises(bnr_amd.BnrError):
        two.run(2, 5, nburn + nsamp + 3)                                      # would write past the table
    for ch in (plain, ring, two):
        ch.close()


@pytest.mark.parametrize("nburn,nsamp,stop", [(7, 5, None), (8, 4, None), (3, 3, None), (9, 4, 6), (2, 3, None)])
def test_smallest_purge_ring(gpu, test1, nburn, nsamp, stop):
    """purge_burn = 1, the smallest ring run! accepts (gibbs.jl:857-860: write row 2, copy it to row 1, read it back from row 1):
    the device ping-pongs between row 2 and a hidden scratch row instead of copying; the table must equal the oracle's (which
    copies like the reference) row for row -- odd and even burn lengths, a run that stops inside the burn-in (rows 1 and 2 then
    both hold the last state, as after copy_table!) and continues in a second call, alone and as a member of a lockstep group.
    With purge_burn = 1 iteration nburn lands in row 2 (not in row purge_burn as for every larger ring), so the samples need
    nsamp + 2 rows: on the nsamp + purge_burn rows initialize_and_run! allocates (gibbs.jl:827-830) the reference stops with a
    BoundsError at the last sample -- here the run is refused up front; the test table therefore has nsamp + 2 rows."""
    X, y = test1
    total = nburn + nsamp
    rows = nsamp + 2
    if nburn > 2:
        short = bnr_amd.Chain(X, y, 5, nsamp + 1, 91, 1)
        short.init_prior()
        with pytest.raises(bnr_amd.BnrError, match="past the table"):
            short.run(2, nburn, total, purge_burn=1)
        short.close()
        os_ = bo.Oracle(X, y, 5, nsamp + 1, 91, pdf_mode=1)
        os_.init_prior()
        with pytest.raises(RuntimeError):
            os_.run(2, nburn, total, purge_burn=1)
    o = bo.Oracle(X, y, 5, rows, 91, pdf_mode=1)
    o.init_prior()
    ch = bnr_amd.Chain(X, y, 5, rows, 91, 1)
    mates = [bnr_amd.Chain.like(ch, 91, c, rows) for c in (2, 3)]
    for c in [ch] + mates:
        c.init_prior()
    grp = bnr_amd.Group([mates[0], ch, mates[1]])
    solo = bnr_amd.Chain.like(ch, 91, 1, rows)
    solo.init_prior()
    if stop is None:
        nxt = o.run(2, nburn, total, purge_burn=1)
        assert solo.run(2, nburn, total, purge_burn=1) == nxt == grp.run(2, nburn, total, purge_burn=1)
    else:
        nxt = o.run(2, nburn, stop, purge_burn=1)
        assert solo.run(2, nburn, stop, purge_burn=1) == nxt == 2 == grp.run(2, nburn, stop, purge_burn=1)
        assert_tables_close(solo.fetch(1, 2), o.t, rows_ref=slice(0, 2), what="stopped inside the burn-in")
        # the reference would continue with run!(..., first_index = 2, ...) on the same table: i restarts at 2
        rest = nburn - stop + 2
        nxt = o.run(2, rest, rest + nsamp, purge_burn=1)
        assert solo.run(2, rest, rest + nsamp, purge_burn=1) == nxt == grp.run(2, rest, rest + nsamp, purge_burn=1)
    assert nxt == nsamp + 3
    last = nxt - 1                                      # rows beyond the last written one were never touched
    a, b = solo.fetch(1, last), ch.fetch(1, last)
    assert_tables_close(a, o.t, rows_ref=slice(0, last), what="purge_burn=1 vs oracle")
    for k in bo.COLUMNS:
        assert np.array_equal(a[k], b[k]), ("group member vs chain alone", k)
    grp.close()
    for c in [ch, solo] + mates:
        c.close()


def _first_rows_match_oracle(X, y, R, seed, rows, members, what):
    """`members` chains of one fit as ONE lockstep group (a single chain when members == 1) for `rows` rows; chain 1 (and one more
    member) against the oracle on identical variates: discrete columns equal, everything else within RTOL."""
    chains = [bnr_amd.Chain(X, y, R, rows, seed, 1)]
    chains += [bnr_amd.Chain.like(chains[0], seed, c, rows) for c in range(2, members + 1)]
    for c in chains:
        c.init_prior()
    grp = bnr_amd.Group(chains) if members > 1 else None
    (grp or chains[0]).run(2, rows, rows)
    for cid in sorted({1, members}):
        o = bo.Oracle(X, y, R, rows, seed, chain=cid, pdf_mode=1)
        o.init_prior()
        o.run(2, rows, rows)
        got = chains[cid - 1].fetch()
        assert_tables_close(got, o.t, what="%s chain %d" % (what, cid))
        assert chains[cid - 1].counters()["chol_fail"] == 0
        _legal_state(got)
    if grp:
        grp.close()
    for c in chains:
        c.close()


def _legal_state(A):
    assert set(np.unique(A["xi"])) <= {0.0, 1.0} and set(np.unique(A["lam"])) <= {0.0, 1.0, -1.0}
    assert np.all(A["S"] > 0) and np.all(A["tau2"] > 0) and np.all(A["theta"] > 0)
    assert np.allclose(A["pi"].sum(axis=2), 1.0)
    for i in range(A["M"].shape[0]):
        M = A["M"][i]
        assert np.allclose(M, M.T, rtol=1e-10) and np.all(np.linalg.eigvalsh(M) > 0)


def test_config3_as_benchmarked_matches_oracle(gpu):
    """BASELINE.json configs[2] exactly as bench.py runs it: n=500, V=100 (q=5050), R=7, the 8 chains as ONE lockstep group
    (gibbs_sample!, gibbs.jl:663-677): five rows of chains 1 and 8 against the oracle."""
    X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
    _first_rows_match_oracle(X, y, 7, 20240501, 5, 8, "cfg3 group of 8")


def test_config4_matches_oracle(gpu):
    """BASELINE.json configs[3]: n=2000, V=200 (q=20100), R=7 -- 63 factorization panels, the blocked trailing update, ksplit for
    long K: the first rows of a chain alone and of a lockstep pair against the oracle (OpenMP Gram on the host cores)."""
    X, y, _ = bnr_amd.make_synthetic(2000, 200, 7, seed=20240501)
    _first_rows_match_oracle(X, y, 7, 20240501, 3, 1, "cfg4 alone")
    _first_rows_match_oracle(X, y, 7, 20240501, 3, 2, "cfg4 pair")


def test_config5_matches_oracle(gpu):
    """BASELINE.json configs[4]: n=500, V=300 (q=45150), R=10 -- the large-q regime (u staged in 24 KB of LDS, 1411 back-projection
    blocks, 300 node workgroups): first rows alone and as a lockstep group of 3 against the oracle."""
    X, y, _ = bnr_amd.make_synthetic(500, 300, 10, seed=20240501)
    _first_rows_match_oracle(X, y, 10, 20240501, 4, 1, "cfg5 alone")
    _first_rows_match_oracle(X, y, 10, 20240501, 3, 3, "cfg5 group of 3")


def test_config_sizes_determinism_and_split_runs(gpu):
    """Size-independent properties at configs[3] and configs[4] (the oracle is too slow for long runs there): two runs of one seed
    are bitwise equal; a run split in two calls (the second call re-derives the carried sums with an explicit X*gamma pass)
    equals the single-call run to 1e-7; every state is legal (discrete values, S > 0, pi rows sum to one, M SPD)."""
    for (n, V, R, tot) in ((2000, 200, 7, 12), (500, 300, 10, 16)):
        X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=20240501)
        a = bnr_amd.Chain(X, y, R, tot, 5, 1)
        b = bnr_amd.Chain.like(a, 5, 1, tot)
        c = bnr_amd.Chain.like(a, 5, 1, tot)
        for ch in (a, b, c):
            ch.init_prior()
        a.run(2, tot, tot)
        b.run(2, tot, tot // 2)
        b.run(tot // 2 + 1, tot, tot)
        c.run(2, tot, tot)
        A, B, Cc = a.fetch(), b.fetch(), c.fetch()
        for k in bo.COLUMNS:
            assert np.array_equal(A[k], Cc[k]), (n, V, k)
            assert np.allclose(A[k], B[k], rtol=1e-7, atol=1e-10), (n, V, k)
        _legal_state(A)
        assert a.counters()["chol_fail"] == 0
        for ch in (a, b, c):
            ch.close()


def test_table_io_move_resize(gpu):
    X, y, _ = bnr_amd.make_synthetic(30, 6, 3, seed=2)
    ch = bnr_amd.Chain(X, y, 3, 10, 5, 1)
    ch.init_prior()
    ch.run(2, 10, 10)
    full = ch.fetch()
    # fetch into a larger host table at an offset, NULL columns skipped, dead columns untouched
    host = bnr_amd.new_table(14, 6, 3, dead=True)
    for k in ("Sigma_inv", "invC", "mu_t"):
        host[k][:] = -7.0
    ch.fetch(3, 8, host, host_row_offset=2)
    for k in bo.COLUMNS:
        assert np.array_equal(host[k][4:10], full[k][2:8]) and np.all(host[k][:4] == 0) and np.all(host[k][10:] == 0)
    assert all(np.all(host[k] == -7.0) for k in ("Sigma_inv", "invC", "mu_t"))
    # copy_table! semantics (utils.jl:72-84), overlapping forward move as the continuation loop does (gibbs.jl:991-993)
    ch.move_rows(1, 5, 6)
    moved = ch.fetch()
    for k in bo.COLUMNS:
        assert np.array_equal(moved[k][:6], full[k][4:10]) and np.array_equal(moved[k][6:], full[k][6:])
    ch.resize(16)
    assert ch.fetch(1, 10)["gamma"].shape[0] == 10 and np.array_equal(ch.fetch(1, 10)["S"], moved["S"])
    # load a reference-layout table back and continue from it: same as the oracle continuing from the same rows
    ch2 = bnr_amd.Chain(X, y, 3, 10, 5, 1)
    ch2.load(full, 1, 4)
    ch2.iter = 4
    ch2.run(5, 10, 10)
    again = ch2.fetch()
    for k in bo.COLUMNS:
        assert np.allclose(again[k], full[k], rtol=1e-9, atol=1e-12), k
    ch.close()
    ch2.close()


def test_chains_on_one_gpu_are_independent_and_async(gpu, test1):
    """Several chains share the GPU (run_async + sync): each equals the chain run alone; chain c uses stream seed+c."""
    X, y = test1
    alone = []
    for c in (1, 2, 3):
        ch = bnr_amd.Chain(X, y, 5, 25, 900, c)
        ch.init_prior()
        ch.run(2, 25, 25)
        alone.append(ch.fetch())
        ch.close()
    chains = [bnr_amd.Chain(X, y, 5, 25, 900, c) for c in (1, 2, 3)]
    for ch in chains:
        ch.init_prior()
    for ch in chains:
        ch.run_async(2, 25, 25)
    for ch in chains:
        assert ch.sync() == 26
    for ch, ref in zip(chains, alone):
        got = ch.fetch()
        for k in bo.COLUMNS:
            assert np.array_equal(got[k], ref[k]), k
        ch.close()
    assert not np.array_equal(alone[0]["gamma"], alone[1]["gamma"])
    o = bo.Oracle(X, y, 5, 25, 900, chain=2, pdf_mode=1)
    o.init_prior()
    o.run(2, 25, 25)
    assert_tables_close(alone[1], o.t, what="chain 2")


def test_lockstep_group_equals_chains_alone(gpu, test1):
    """bnr_group_run: several chains of one fit on one GPU advance together (one launch per kernel, blockIdx.z = chain).
    Every member's table must be bitwise what the chain produces alone -- over the purge ring, a continuation call, all
    launch modes, a progress callback, and a ragged size whose n is not a multiple of the tile sizes."""
    X, y = test1
    Xr, yr, _ = bnr_amd.make_synthetic(37, 9, 3, seed=5)
    # (X, y, R, table length, [run! calls (first_index, nburn, total, purge_burn)])
    cases = [(X, y, 5, 14, [(2, 20, 30, 4)]),                       # purge ring: 30 iterations in a 14-row table
             (X, y, 5, 30, [(2, 12, 9, None), (10, 12, 30, None)]),  # continuation (run! with first_index > 2)
             (Xr, yr, 3, 21, [(2, 5, 9, None), (10, 5, 21, None)])]
    for Xc, yc, R, tot, calls in cases:
        alone = []
        for c in (1, 2, 3, 4):
            ch = bnr_amd.Chain(Xc, yc, R, tot, 77, c)
            ch.init_prior()
            for call in calls:
                ch.run(*call)
            alone.append(ch.fetch())
            ch.close()
        assert not np.array_equal(alone[0]["gamma"], alone[1]["gamma"])
        for opts in ({}, {"graph": 0}, {"overlap": 0, "graph_k": 3}):
            chains = [bnr_amd.Chain(Xc, yc, R, tot, 77, 1)]
            chains += [bnr_amd.Chain.like(chains[0], 77, c) for c in (2, 3, 4)]        # X, y shared on the device
            for ch in chains:
                ch.init_prior()
            g = bnr_amd.Group(chains)
            for k, v in opts.items():
                g.set_option(k, v)
            ticks = []
            for i, call in enumerate(calls):
                g.run(*call, prog_freq=4, callback=(lambda done: ticks.append(done)) if i == 0 else None)
            first = calls[0]
            assert ticks == [d for d in range(1, first[2] - first[0] + 2) if (first[0] + d - 1) % 4 == 0]   # gibbs.jl:854-856
            for ch, ref in zip(chains, alone):
                got = ch.fetch()
                for k in bo.COLUMNS:
                    assert np.array_equal(got[k], ref[k]), (opts, k)
                assert ch.counters()["chol_fail"] == 0
            g.close()
            for ch in chains:
                ch.close()
    a, b = bnr_amd.Chain(X, y, 5, 10, 1, 1), bnr_amd.Chain(Xr, yr, 3, 10, 1, 2)
    with pytest.raises(bnr_amd.BnrError):
        bnr_amd.Group([a, b])                                        # unequal shapes
    # the shared inputs outlive the chain that uploaded them
    c2 = bnr_amd.Chain.like(a, 1, 2)
    a.close()
    c2.init_prior()
    c2.run(2, 10, 10)
    ref = bnr_amd.Chain(X, y, 5, 10, 1, 2)
    ref.init_prior()
    ref.run(2, 10, 10)
    assert np.array_equal(c2.fetch()["gamma"], ref.fetch()["gamma"])
    for ch in (b, c2, ref):
        ch.close()


def test_handles_can_be_driven_from_different_host_threads(gpu, test1):
    """ABI contract (threading): a handle is not shared between threads, but different handles may be driven
    concurrently from different host threads (the reference runs one chain per pmap worker).  Two groups and a single
    chain run at the same time -- stream captures included -- and every table equals the chain run alone."""
    import threading
    X, y = test1
    tot = 40
    alone = {}
    for c in range(1, 6):
        ch = bnr_amd.Chain(X, y, 5, tot, 5, c)
        ch.init_prior()
        ch.run(2, tot, tot)
        alone[c] = ch.fetch()
        ch.close()
    first = bnr_amd.Chain(X, y, 5, tot, 5, 1)
    chains = {1: first}
    for c in range(2, 6):
        chains[c] = bnr_amd.Chain.like(first, 5, c)
    for ch in chains.values():
        ch.init_prior()
    runners = [bnr_amd.Group([chains[1], chains[2]]), bnr_amd.Group([chains[3], chains[4]]), chains[5]]
    errors = []

    def work(r):
        try:
            r.run(2, tot, tot)
        except Exception as e:                                   # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(r,)) for r in runners]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for c, ch in chains.items():
        got = ch.fetch()
        for k in bo.COLUMNS:
            assert np.array_equal(got[k], alone[c][k]), (c, k)
    for r in runners[:2]:
        r.close()
    for ch in chains.values():
        ch.close()


def test_generate_samples_end_to_end(gpu, test1, tmp_path):
    """generate_samples! / Fit! drop-in (gibbs.jl:725-751, 897-1020): Results layout, Rhat over 2 chains, top-up loop."""
    X, y = test1
    keep = []
    res = bnr_amd.generate_samples(X, y, 5, nburn=40, nsamp=30, maxburn=40, psrf_cutoff=1.2, x_transform=False,
                                   suppress_timer=True, num_chains=2, seed=1234, _keep=keep)
    assert res.burn_in == 40 and res.sampled == 30
    assert res.state["gamma"].shape == (70, 190, 1) and res.state["u"].shape == (70, 5, 19) and res.state["pi"].shape == (70, 5, 3)
    assert set(res.state) == set(bnr_amd._capi.TABLE_COLUMNS + bnr_amd._capi.DEAD_COLUMNS)
    cs = keep[0]
    tabs = [cs.chains[c].fetch() for c in (1, 2)]
    both_g = np.stack([t["gamma"][40:70, :, 0] for t in tabs], axis=2)
    both_x = np.stack([t["xi"][40:70, :, 0] for t in tabs], axis=2)
    assert np.allclose(res.rhatgamma, bo.rhat(both_g), rtol=1e-10) and np.allclose(res.rhatxi, bo.rhat(both_x), rtol=1e-10)
    assert np.array_equal(res.state["gamma"], tabs[0]["gamma"])               # only chain 1's trace is returned (gibbs.jl:788)
    o = bo.Oracle(X, y, 5, 70, 1234, chain=1, pdf_mode=1)
    o.init_prior()
    o.run(2, 40, 70)
    assert_tables_close(tabs[0], o.t, what="generate_samples chain 1")
    cs.close()
    # top-up loop: an impossible cutoff forces exactly one extra round of nburn samples (maxburn = nburn + 1)
    keep = []
    res2 = bnr_amd.generate_samples(X, y, 5, nburn=20, nsamp=10, maxburn=21, psrf_cutoff=0.5, x_transform=False,
                                    suppress_timer=True, num_chains=2, seed=5, _keep=keep)
    o = bo.Oracle(X, y, 5, 30, 5, chain=1, pdf_mode=1)
    o.init_prior()
    o.run(2, 20, 30)
    for i in range(10):                                                       # copy_table! loop, gibbs.jl:991-993
        for k in bo.COLUMNS:
            o.t[k][i] = o.t[k][20 + i]
    o.run(11, 20, 30)                                                         # run!(…, num2move+1, nburn-nsamp+num2move, num2move+nburn)
    assert_tables_close(keep[0].chains[1].fetch(), o.t, what="top-up round")
    keep[0].close()
    # Fit! writes parameters.log and dispatches on mingen/maxgen (doubling scheme)
    log = tmp_path / "parameters.log"
    r3 = bnr_amd.Fit(X, y, 5, mingen=20, maxgen=40, psrf_cutoff=0.5, x_transform=False, suppress_timer=True,
                     num_chains=2, seed=9, filename=str(log))
    txt = log.read_text()
    assert "BayesianNetworkRegression.jl Fit! function" in txt and "seed=9" in txt and "mingen=20, maxgen=40" in txt
    assert r3.burn_in == 10 and r3.sampled == 20 and r3.state["gamma"].shape[0] == 30


def _oracle_doubling(X, y, R, mingen, rounds, seed, chain):
    """generate_samples_dbl! (gibbs.jl:1051-1198) walked by the oracle for ONE chain: first pass nburn = round(mingen/2),
    nsamp = mingen - nburn; then `rounds` doubling rounds -- a NEW table of tot_samples + halfburn rows, the last num2move rows
    of the old one copied to its head (copy_table!, :1172), run! from num2move + 1 with nburn = 0 (:1176-1178)."""
    nburn = int(round(mingen / 2))
    nsamp = mingen - nburn
    tot_save = nburn + nsamp
    o = bo.Oracle(X, y, R, tot_save, seed, chain=chain, pdf_mode=1)
    o.init_prior()
    o.run(2, nburn, tot_save)
    tot_samples = nsamp
    for _ in range(rounds):
        halfburn = int(round(mingen / 2))
        num2move = tot_samples
        tot_samples += halfburn
        nsamp = tot_samples
        tot_sze, tot_save = tot_save, tot_samples + halfburn
        new = bo.new_table(tot_save, o.V, R)
        for k in bo.COLUMNS:
            new[k][:num2move] = o.t[k][tot_sze - num2move:tot_sze]
        it = o.iter
        o = bo.Oracle(X, y, R, tot_save, seed, chain=chain, pdf_mode=1, table=new)
        o.iter = it
        assert o.run(num2move + 1, 0, tot_save) == tot_save + 1
    return o.t, nburn, nsamp


def test_doubling_scheme_walks_the_same_rows_as_the_oracle(gpu, test1, capfd):
    """generate_samples_dbl! (gibbs.jl:1051-1198; Fit! with mingen, maxgen): an unreachable PSRF cutoff forces exactly two doubling
    rounds (maxgen = 3 mingen); chain 1's final table, burn_in / sampled and the PSRF over both chains equal what the oracle
    produces walking the reference's row-moving arithmetic (num2move, re-allocated table, first_index) -- also with an odd mingen."""
    X, y = test1
    for mingen in (24, 15):
        res = bnr_amd.generate_samples_dbl(X, y, 5, mingen=mingen, maxgen=3 * mingen, psrf_cutoff=0.5, x_transform=False,
                                           suppress_timer=True, num_chains=2, seed=321)
        err = capfd.readouterr().err
        tabs = [_oracle_doubling(X, y, 5, mingen, 2, 321, c) for c in (1, 2)]
        t1, nburn, nsamp = tabs[0]
        assert res.burn_in == nburn and res.sampled == nsamp and res.state["gamma"].shape[0] == nburn + nsamp == t1["gamma"].shape[0]
        assert_tables_close(res.state, t1, what="doubling scheme, mingen=%d" % mingen)
        g = np.stack([t[0]["gamma"][nburn:nburn + nsamp, :, 0] for t in tabs], axis=2)
        x = np.stack([t[0]["xi"][nburn:nburn + nsamp, :, 0] for t in tabs], axis=2)
        assert np.allclose(res.rhatgamma, bo.rhat(g), rtol=1e-6) and np.allclose(res.rhatxi, bo.rhat(x), rtol=1e-6, equal_nan=True)
        hb = int(round(mingen / 2))
        first = mingen - hb
        assert "num2move: %d nburn: %d nsamp: %d tot_save: %d first_index: %d" % (first, hb, first + hb, first + 2 * hb, first + 1) in err
        assert "%d samples generated" % (3 * mingen) in err


def test_input_formats_are_converted_on_the_device(gpu):
    """The inputs generate_samples! accepts (gibbs.jl:907-918, docs/src/man/inputdata.md:5-10): X_new keeps the element type of the
    data (Matrix{eltype(T)}: Bool for 0/1 adjacency matrices, Int counts, Float32/64), and with x_transform=true X is the vector of
    adjacency matrices that setup_X! (gibbs.jl:239-247) vectorises by lower_triangle (utils.jl:40-57: A[l, k], l >= k, so a matrix
    that is only lower-triangular gives the same row).  Here the raw bytes are uploaded and converted / vectorised on the device:
    every variant must give the table of the float64 n x q matrix bit for bit."""
    rng = np.random.default_rng(5)
    n, V, R, tot = 37, 9, 3, 8
    mats_bool = [np.tril(rng.random((V, V)) < 0.5) for _ in range(n)]                  # lower-triangular Bool, like examples/data*.csv
    mats_int = [np.tril(rng.integers(0, 6, (V, V))) + np.triu(rng.integers(0, 6, (V, V)), 1) for _ in range(n)]   # not symmetric
    y = rng.normal(size=n)

    def table(X, x_transform, bytes_expected=None):
        ch = bnr_amd.Chain(bnr_amd.XInput(X, x_transform), y, R, tot, 11, 1)
        if bytes_expected is not None:                                                  # 0..255-valued integer input keeps a byte image of X
            assert (ch.last_timing(3)[1] == 1) == bytes_expected, (np.asarray(X[0]).dtype, x_transform)
        # a 0/1 integer-typed matrix gets its Gram from the i8 matrix pipe by default (tested on its own below: close, not bitwise);
        # THIS test is about the conversions, so every variant runs the f64 Gram
        binary = bytes_expected is True and np.isin(np.asarray(bnr_amd.setup_X(X, True)[0] if x_transform else X), (0, 1)).all()
        assert ch.last_timing(4) == (0.0, 0)                                           # (a problem of this size defaults to the f64 Gram)
        if binary:
            ch.set_option("gram_i8", 1)
            assert ch.last_timing(4)[0] == 1 and ch.last_timing(4)[1] in (7, 8)
            ch.set_option("gram_i8", 0)
            assert ch.last_timing(4) == (0.0, 0)
        else:
            with pytest.raises(bnr_amd.BnrError):
                ch.set_option("gram_i8", 1)
        ch.init_prior()
        ch.run(2, tot, tot)
        t = ch.fetch()
        ch.close()
        return t

    for mats in (mats_bool, mats_int):
        Xh, Vh, qh = bnr_amd.setup_X(mats, True)                                      # host restatement of setup_X!
        assert (Vh, qh) == (V, V * (V + 1) // 2)
        want = table(Xh, False)
        variants = {"matrices": (mats, True), "matrices f32": ([m.astype(np.float32) for m in mats], True),
                    "matrix own dtype": (Xh.astype(mats[0].dtype), False), "matrix int32": (Xh.astype(np.int32), False),
                    "matrix uint8": (Xh.astype(np.uint8), False), "matrix f32": (Xh.astype(np.float32), False),
                    "matrix int16 (promoted on the host)": (Xh.astype(np.int16), False)}
        for name, (X, tr) in variants.items():
            narrow = np.asarray(X[0]).dtype in (np.dtype(bool), np.dtype(np.uint8), np.dtype(np.int32), np.dtype(np.int64))
            got = table(X, tr, bytes_expected=narrow)
            for k in bo.COLUMNS:
                assert np.array_equal(got[k], want[k]), (name, k)
    # an integer matrix with an entry outside 0..255 keeps no byte image (and still gives the f64 table)
    Xbig = bnr_amd.setup_X(mats_int, True)[0].astype(np.int64)
    Xbig[3, 4] = 256
    wantb = table(Xbig.astype(np.float64), False, bytes_expected=False)
    gotb = table(Xbig, False, bytes_expected=False)
    for k in bo.COLUMNS:
        assert np.array_equal(gotb[k], wantb[k]), ("int64 with 256", k)
    o = bo.Oracle(bnr_amd.setup_X(mats_int, True)[0], y, R, tot, 11, chain=1, pdf_mode=1)
    o.init_prior()
    o.run(2, tot, tot)
    assert_tables_close(table(mats_int, True), o.t, what="integer adjacency matrices vs oracle")
    # the drop-in entry point takes the vector of matrices as the reference does (x_transform = true is its default)
    res = bnr_amd.generate_samples(mats_bool, y, R, nburn=6, nsamp=4, maxburn=6, psrf_cutoff=10.0, suppress_timer=True, num_chains=2, seed=10)
    assert res.state["gamma"].shape == (10, V * (V + 1) // 2, 1)
    with pytest.raises(ValueError):
        bnr_amd.XInput([np.zeros((3, 3)), np.zeros((4, 4))], True)


def test_byte_image_of_a_binary_model_matrix_at_config5_size(gpu):
    """SURVEY 8f-2 (gibbs.jl:239-247, 907-918; docs/src/man/inputdata.md:5-10: adjacency data is Bool): at BASELINE configs[4]'s size
    (n=500, V=300, q=45 150: X = 180.6 MB as Float64, 22.6 MB as bytes) a 0/1 model matrix stays in HBM as bytes for the two bandwidth-bound
    passes over X (k_xpass, k_backproj); the tables are bitwise those of the same chain reading the f64 image (option byte_x = 0), of the
    same data given as float64, alone and as members of a lockstep group."""
    n, V, R, tot = 500, 300, 10, 4
    rng = np.random.default_rng(9)
    q = V * (V + 1) // 2
    Xb = rng.random((n, q)) < 0.5
    y = rng.normal(size=n)
    want = None
    for name, X, byte_x in (("float64", Xb.astype(np.float64), None), ("bool, bytes", Xb, 1), ("bool, f64 image", Xb, 0)):
        ch = bnr_amd.Chain(bnr_amd.XInput(np.asfortranarray(X), False), y, R, tot, 21, 1)
        mate = bnr_amd.Chain.like(ch, 21, 2, tot)
        if byte_x is not None:
            assert ch.last_timing(3)[1] == 1
            for c in (ch, mate):
                c.set_option("byte_x", byte_x)
                c.set_option("gram_i8", 0)                  # (the i8 Gram of a binary matrix is close, not bitwise: its own test below)
            assert ch.last_timing(3)[1] == byte_x
        else:
            assert ch.last_timing(3)[1] == 0
        for c in (ch, mate):
            c.init_prior()
        g = bnr_amd.Group([ch, mate])
        g.run(2, tot, tot)
        solo = bnr_amd.Chain.like(ch, 21, 1, tot)
        if byte_x is not None:
            solo.set_option("byte_x", byte_x)
            solo.set_option("gram_i8", 0)
        solo.init_prior()
        solo.run(2, tot, tot)
        tabs = (ch.fetch(), solo.fetch())
        assert ch.counters()["chol_fail"] == 0
        g.close()
        for c in (ch, mate, solo):
            c.close()
        for k in bo.COLUMNS:
            assert np.isfinite(tabs[0][k]).all()
            assert np.array_equal(tabs[0][k], tabs[1][k]), (name, "group vs alone", k)
            if want is not None:
                assert np.array_equal(tabs[0][k], want[k]), (name, k)
        want = want or tabs[0]
    # ... and they are the ORACLE's tables on the same 0/1 data (round 3 compared the library with itself here): the prior draw and two sweeps
    o = bo.Oracle(np.asfortranarray(Xb.astype(np.float64)), y, R, 3, 21, chain=1, pdf_mode=1)
    o.init_prior()
    o.run(2, 3, 3)
    assert_tables_close({k: v[:3] for k, v in want.items()}, o.t, what="0/1 model matrix at config-5 size (byte image) vs oracle")


@pytest.mark.parametrize("n,V,R,nmates", [(500, 300, 10, 2), (500, 100, 7, 7), (70, 19, 5, 1), (130, 40, 3, 0), (2000, 200, 7, 0)])
def test_binary_model_matrix_gram_on_the_i8_matrix_pipe(gpu, n, V, R, nmates):
    """SURVEY 8f-2, second half (docs/src/man/inputdata.md:5-10: the inputs are 0/1 adjacency data; X_new = Matrix{eltype(T)} gibbs.jl:917;
    the Gram Xtau tau2 D Xtau' of gibbs.jl:434): a Bool model matrix gets its Gram from v_mfma_i32_16x16x64_i8 -- S cut into i8L planes of
    balanced base-256 digits under a common exponent, one exact i32 Gram per plane, recombined in f64 (k_sdigits, k_gram_i8).  At BASELINE configs[4]'s
    and configs[2]'s sizes, a small one and one whose n is no multiple of 64:
      * G of the i8 path against G of the f64 path (same S): |dG| <= 1e-12 max |G|, the stated bound (the f64 path's own rounding);
      * the tables against the ORACLE on the same 0/1 data: rtol 1e-6 like every other path;  against the f64-Gram tables: 1e-8;
      * members of a lockstep group = the chain alone (bitwise: per layout the arithmetic is the same)."""
    tot = 4
    rng = np.random.default_rng(31 + n + V)
    q = V * (V + 1) // 2
    Xb = np.asfortranarray(rng.random((n, q)) < 0.5)
    y = rng.normal(size=n)
    tabs, grams = {}, {}
    for mode in ("i8", "f64"):
        ch = bnr_amd.Chain(bnr_amd.XInput(Xb, False), y, R, tot, 77, 1)
        mates = [bnr_amd.Chain.like(ch, 77, 2 + m, tot) for m in range(nmates)]
        dm = ch.debug_dims()
        assert (ch.last_timing(4)[0] == 1) == (dm["n_pad"] ** 2 * q >= 2.5e8)            # the default: on where it was measured faster (profiles/round5_gram_i8.txt)
        for c in [ch] + mates:
            c.set_option("gram_i8", 1 if mode == "i8" else 0)
        assert ch.last_timing(4) == ((1.0, dm["i8L"]) if mode == "i8" else (0.0, 0))
        for c in [ch] + mates:
            c.init_prior()
        # ONE sweep alone first: the Gram of sweep 1 sees the prior draw's S in both modes
        solo = bnr_amd.Chain.like(ch, 77, 1, tot)
        solo.set_option("gram_i8", 1 if mode == "i8" else 0)
        solo.init_prior()
        solo.run(2, tot, 2)
        grams[mode] = solo.debug_gram()
        solo.run(3, tot, tot)
        if nmates:
            g = bnr_amd.Group([ch] + mates)
            assert g.last_timing(4)[0] == (1 if mode == "i8" else 0)
            g.run(2, tot, tot)
            t = ch.fetch()
            one = bnr_amd.Chain.like(ch, 77, 1, tot)
            one.set_option("gram_i8", 1 if mode == "i8" else 0)
            one.init_prior()
            one.run(2, tot, tot)
            ts = one.fetch()
            one.close()
            for k in bo.COLUMNS:
                assert np.array_equal(t[k], ts[k]), (mode, "group member vs alone", k)
            g.close()
        tabs[mode] = solo.fetch()
        assert solo.counters()["chol_fail"] == 0
        for c in [ch, solo] + mates:
            c.close()
    gmax = np.abs(grams["f64"]).max()
    err = np.abs(grams["i8"] - grams["f64"])[:n, :n].max() / gmax
    assert err <= 1e-12, ("i8 Gram vs f64 Gram, relative to max |G|", err)
    assert gmax > 0 and np.isfinite(grams["i8"]).all()
    for k in bo.COLUMNS:
        assert np.allclose(tabs["i8"][k], tabs["f64"][k], rtol=1e-8, atol=1e-11), ("i8-Gram tables vs f64-Gram tables", k, np.abs(tabs["i8"][k] - tabs["f64"][k]).max())
    if n >= 1000:
        return              # BASELINE configs[3]'s size (63 panels, two-panel factorization, i8L = 8): Gram bound and table closeness only -- the oracle takes minutes here
    o = bo.Oracle(np.asfortranarray(Xb.astype(np.float64)), y, R, tot, 77, chain=1, pdf_mode=1)
    o.init_prior()
    o.run(2, tot, tot)
    assert_tables_close(tabs["i8"], o.t, what="binary model matrix, Gram on the i8 pipe, vs oracle (n=%d V=%d)" % (n, V))


def test_device_summary_equals_host_summary(gpu, test1):
    """bnr_chain_summary (Summary on the device, gibbs.jl:1214-1250): the posterior means agree with the host to rounding,
    the order statistics are EXACTLY the entries of the sorted trace the reference indexes -- also with ties, negative
    values, +-0 and a window that does not start at row 1 -- and Summary() prints the same table from either source."""
    X, y = test1
    tot = 120
    ch = bnr_amd.Chain(X, y, 5, tot, 2024, 1)
    ch.init_prior()
    ch.run(2, tot, tot)
    t = ch.fetch()
    # plant ties / signed zeros / extreme values in a few gamma columns (the kernel only reads the table)
    t["gamma"][40:, 0, 0] = 0.25
    t["gamma"][40:80, 1, 0] = -0.0
    t["gamma"][80:, 1, 0] = 0.0
    t["gamma"][40:, 2, 0] = np.where(np.arange(80) % 2 == 0, -1e300, 1e-300)      # extreme exponents (selection works on the bit image)
    t["gamma"][40:, 3, 0] = -np.abs(t["gamma"][40:, 3, 0])
    ch.load(t)
    for nburn, nsamp, interval in ((40, 80, 95), (0, 120, 95), (17, 101, 50), (60, 60, 90)):
        g = t["gamma"][nburn:nburn + nsamp, :, 0]
        gs = np.sort(g, axis=0)
        lw, hi = bnr_amd.api._summary_ranks(nsamp, interval)
        dev = bnr_amd.device_summary(ch, nburn, nsamp, interval)
        assert np.array_equal(dev["lower_bound"], gs[lw - 1]) and np.array_equal(dev["upper_bound"], gs[hi - 1])
        assert np.all(np.abs(dev["estimate"] - g.mean(axis=0)) <= 1e-13 * np.abs(g).mean(axis=0))     # summation order only
        assert np.allclose(dev["probability"], t["xi"][nburn:nburn + nsamp, :, 0].mean(axis=0), rtol=1e-13)
        full = bnr_amd.new_table(tot, 19, 5, dead=True)
        ch.fetch(1, tot, full)
        host = bnr_amd.Summary(bnr_amd.Results(full, None, None, nburn, nsamp), interval)
        devs = bnr_amd.Summary(bnr_amd.Results(None, None, None, nburn, nsamp, dev), interval)
        for k in ("lower_bound", "upper_bound"):
            assert np.array_equal(host.edge_coef[k], devs.edge_coef[k]), k
        # the means are sums in a different order: equal to rounding, so the 3-digit table can differ by one unit in the
        # last printed digit only where a mean sits on a rounding boundary (and not at all for moderate magnitudes)
        ok = np.abs(host.edge_coef["estimate"]) < 1e6
        assert np.allclose(host.edge_coef["estimate"][ok], devs.edge_coef["estimate"][ok], rtol=0, atol=1.0001e-3)
        assert np.mean(host.edge_coef["estimate"][ok] == devs.edge_coef["estimate"][ok]) > 0.98
        assert np.allclose(host.prob_nodes["probability"], devs.prob_nodes["probability"], rtol=0, atol=1.0001e-3)
    with pytest.raises(IndexError):
        bnr_amd.device_summary(ch, 60, 60, 99)                      # round(60 * 0.005) = 0: the reference stops with a BoundsError
    with pytest.raises(bnr_amd.BnrError):
        ch.summary(1, tot, 0, 5)
    with pytest.raises(bnr_amd.BnrError):
        ch.summary(100, 50, 1, 5)
    ch.close()
    # through the fit: Results without the state table, Summary from the device statistics
    r = bnr_amd.Fit(X, y, 5, nburn=30, nsamples=40, psrf_cutoff=50.0, x_transform=False, suppress_timer=True, num_chains=2,
                    seed=77, filename=None, return_state=False, summary_interval=95)
    r2 = bnr_amd.Fit(X, y, 5, nburn=30, nsamples=40, psrf_cutoff=50.0, x_transform=False, suppress_timer=True, num_chains=2,
                     seed=77, filename=None)
    assert r.state is None and r.summary_device is not None and r2.summary_device is None
    sa, sb = bnr_amd.Summary(r), bnr_amd.Summary(r2)
    assert np.array_equal(sa.edge_coef["lower_bound"], sb.edge_coef["lower_bound"]) and np.array_equal(sa.edge_coef["upper_bound"], sb.edge_coef["upper_bound"])
    assert np.allclose(sa.edge_coef["estimate"], sb.edge_coef["estimate"], rtol=0, atol=1.0001e-3)


def test_posterior_recovers_synthetic_truth(gpu):
    """End to end on data drawn from the model itself (SURVEY 8d generator: n=200, V=20, R=5): 8 chains x 10 000 iterations
    as one lockstep group converge (split-Rhat < 1.1 for every gamma and xi), the posterior mean of gamma reproduces the
    true edge coefficients, the 95 % intervals cover them, and P(xi = 1) separates the influential nodes from the rest."""
    n, V, R, nburn, nsamp, C = 200, 20, 5, 5000, 5000, 8
    X, y, truth = bnr_amd.make_synthetic(n, V, R, seed=3)
    tot = nburn + nsamp
    chains = [bnr_amd.Chain(X, y, R, tot, 99, 1)]
    chains += [bnr_amd.Chain.like(chains[0], 99, c, tot) for c in range(2, C + 1)]
    for ch in chains:
        ch.init_prior()
    g = bnr_amd.Group(chains)
    g.run(2, nburn, tot)
    q = V * (V + 1) // 2
    rh = bnr_amd.rhat_from_stats(np.stack([ch.rhat_stats(nburn + 1, nsamp) for ch in chains]), nsamp)
    assert np.nanmax(rh[:q]) < 1.1 and np.nanmax(rh[q:]) < 1.1
    lw, hi = bnr_amd.api._summary_ranks(nsamp, 95)
    summ = [ch.summary(nburn + 1, nsamp, lw, hi) for ch in chains]
    mean = np.mean([s[0] for s in summ], axis=0)
    lo, up = np.mean([s[1] for s in summ], axis=0), np.mean([s[2] for s in summ], axis=0)
    pxi = np.mean([s[3] for s in summ], axis=0)
    B = truth["B"]
    assert np.corrcoef(mean, B)[0, 1] > 0.99
    assert np.sqrt(np.mean((mean - B) ** 2)) < 0.15 * np.sqrt(np.mean(B ** 2))
    assert np.mean((B >= lo) & (B <= up)) >= 0.93
    assert np.all(pxi[truth["xi"] == 1] > 0.9) and np.all(pxi[truth["xi"] == 0] < 0.1)
    for ch in chains:
        c = ch.counters()
        assert c["chol_fail"] == 0 and c["nan_w"] == 0 and c["sampler_cap"] == 0
    g.close()
    for ch in chains:
        ch.close()


def test_reference_example_data_through_the_drop_in_api(gpu):
    """BASELINE.json configs[0]: the reference's example (examples/matrix_networks.csv + responses.csv: n=100, V=30, R=5)
    through Fit! with the Summary statistics computed on the device, against the truth the example was simulated from
    (examples/true_b.csv, true_xi.csv): the 20 influential nodes are found with probability ~1, no other node gets more than
    0.5, and the posterior means of the 435 off-diagonal edge coefficients follow true_b."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "examples_xy.npz"))
    X, y, tb, txi = d["X"], d["y"], d["true_b"], d["true_xi"]
    res = bnr_amd.Fit(X, y, 5, nburn=4000, nsamples=4000, num_chains=4, seed=1234, x_transform=False, suppress_timer=True,
                      filename=None, psrf_cutoff=1.3, return_state=False, summary_interval=95)
    assert res.state is None and res.sampled == 4000
    s = bnr_amd.Summary(res)
    off = s.edge_coef["node1"] != s.edge_coef["node2"]
    assert off.sum() == 435
    est = res.summary_device["estimate"][off]
    assert np.corrcoef(est, tb)[0, 1] > 0.75
    p = res.summary_device["probability"]
    assert np.all(p[txi == 1] > 0.9) and np.all(p[txi == 0] < 0.5)
    lo, hi = res.summary_device["lower_bound"][off], res.summary_device["upper_bound"][off]
    assert np.mean((tb >= lo) & (tb <= hi)) > 0.8
    assert len(str(s).splitlines()) == 3 + 465 + 1 + 30


def test_ess_message_and_fit_option(gpu, test1):
    """bnr_chain_ess_stats (device autocovariances, an addition to the reference): equal to numpy on the fetched table;
    ChainSet.ess / Fit(ess_max_lag=...) give finite effective sample sizes no larger than N log10 N."""
    from test_host_cpu import _ess_message
    X, y = test1
    tot, first, nsamp, L = 400, 101, 300, 40
    ch = bnr_amd.Chain(X, y, 5, tot, 11, 1)
    ch.init_prior()
    ch.run(2, tot, tot)
    t = ch.fetch()
    msg = ch.ess_stats(first, nsamp, L)
    par = np.concatenate([t["gamma"][first - 1:first - 1 + nsamp, :, 0], t["xi"][first - 1:first - 1 + nsamp, :, 0]], axis=1)
    ref = _ess_message(par, L)
    assert msg.shape == ref.shape
    scale = np.abs(ref).max()
    assert np.allclose(msg, ref, rtol=1e-9, atol=1e-12 * scale)
    with pytest.raises(bnr_amd.BnrError):
        ch.ess_stats(first, nsamp, nsamp)                       # max_lag > nsamp / 2
    ch.close()
    r = bnr_amd.Fit(X, y, 5, nburn=300, nsamples=400, psrf_cutoff=50.0, x_transform=False, suppress_timer=True, num_chains=3,
                    seed=5, filename=None, ess_max_lag=0)
    n_total = 3 * 400
    assert r.essgamma.shape == (190,) and r.essxi.shape == (19,)
    g = r.essgamma[np.isfinite(r.essgamma)]
    assert g.size > 150 and np.all(g > 1) and np.all(g <= n_total * np.log10(n_total) + 1e-6)


def test_random_shapes_and_hyperparameters_match_oracle(gpu):
    """Randomised parity sweep (a fixed-seed slice of tools/fuzz_shapes.py, which was run over 200 cases): random n (incl.
    the tile-size boundaries 63/64/65, 127/128/129 and n = 1..3), V, R, X flavour and hyper-parameters (incl. aDelta = 0 /
    bDelta = 0, eta < 1), the chain alone or as the middle member of a lockstep group, against the oracle on identical
    variates: discrete columns equal, everything else within 1e-6."""
    rng = np.random.default_rng(12345)
    for case in range(24):
        V = int(rng.integers(2, 33)); R = int(rng.integers(1, 13))
        n = int(rng.choice([1, 2, 3, 31, 33, 63, 64, 65, 100, 127, 128, 129, int(rng.integers(4, 200))]))
        tot = int(rng.integers(3, 8)); seed = int(rng.integers(1, 10**6)); normal_x = bool(rng.integers(0, 2)); group = bool(rng.integers(0, 2))
        hyper = dict(eta=float(rng.choice([1.01, 0.5, 2.0])), zeta=float(rng.choice([1.0, 0.3])), iota=float(rng.choice([1.0, 2.5])),
                     aDelta=float(rng.choice([1.0, 0.0, 3.0])), bDelta=float(rng.choice([1.0, 0.0, 2.0])), nu=float(max(R, rng.choice([10, 12, R + 1]))))
        X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=seed, normal_x=normal_x)
        ch = bnr_amd.Chain(X, y, R, tot, seed, 1, **hyper)
        mates = [bnr_amd.Chain.like(ch, seed, c, tot) for c in (2, 3)] if group else []
        for c in [ch] + mates:
            c.init_prior()
        g = bnr_amd.Group([mates[0], ch, mates[1]]) if group else None
        (g or ch).run(2, tot, tot)
        o = bo.Oracle(X, y, R, tot, seed, chain=1, pdf_mode=1, **hyper)
        o.init_prior()
        o.run(2, tot, tot)
        assert_tables_close(ch.fetch(), o.t, what="case %d: n=%d V=%d R=%d group=%s %s" % (case, n, V, R, group, hyper))
        assert ch.counters()["chol_fail"] == 0
        if g:
            g.close()
        for c in [ch] + mates:
            c.close()


def test_shapes_at_the_lds_budgets(gpu):
    """u (R x V) and the n-vectors are staged in LDS by single workgroups: shapes that need more than the default 64 KiB of
    dynamic LDS run (R V = 9600: 75 KiB in k_tail; R = 32 is the latent-dimension limit); n beyond the LDS of the CU is
    rejected at bnr_chain_create with a message instead of failing at a launch; R V beyond it runs with u read from the row."""
    X, y, _ = bnr_amd.make_synthetic(20, 300, 32, seed=1)
    ch = bnr_amd.Chain(X, y, 32, 4, 1, 1, nu=34)                # the inverse Wishart needs nu >= R
    ch.init_prior()
    ch.run(2, 4, 4)
    t = ch.fetch()
    assert all(np.all(np.isfinite(t[k])) for k in bo.COLUMNS)
    c = ch.counters()
    assert c["chol_fail"] == 0 and c["nan_w"] == 0
    ch.close()
    # R V beyond what k_tail can stage (15 360 doubles): u is read from the trace row instead -- the reference has no such limit (round 5; rounds 1-4 refused
    # the shape).  R V = 19 200, q = 180 300: the prior draw and two sweeps against the oracle, alone and in a lockstep group
    Xb, yb, _ = bnr_amd.make_synthetic(5, 600, 32, seed=1)
    chb, ob = pair(Xb, yb, 32, 3, 1, nu=34)
    mate = bnr_amd.Chain.like(chb, 1, 2, 3)
    chb.init_prior(); mate.init_prior(); ob.init_prior()
    g = bnr_amd.Group([chb, mate])
    g.run(2, 3, 3)
    ob.run(2, 3, 3)
    assert_tables_close(chb.fetch(), ob.t, what="R V = 19200 (u from the row)")
    solo = bnr_amd.Chain.like(chb, 1, 1, 3)
    solo.init_prior()
    solo.run(2, 3, 3)
    tg, ts = chb.fetch(), solo.fetch()
    assert all(np.array_equal(tg[k], ts[k]) for k in bo.COLUMNS)
    g.close()
    for c_ in (chb, mate, solo):
        c_.close()
    with pytest.raises(bnr_amd.BnrError, match="n must not exceed"):
        bnr_amd.Chain(np.zeros((14001, 3), order="F"), np.zeros(14001), 2, 4, 1, 1)


def test_bad_arguments_are_reported(gpu):
    X, y, _ = bnr_amd.make_synthetic(8, 4, 2, seed=1)
    with pytest.raises(bnr_amd.BnrError) as e:
        bnr_amd.Chain(X, y, 40, 4, 1, 1)
    assert e.value.code == 1
    with pytest.raises(bnr_amd.BnrError):
        bnr_amd.Chain(X, y, 2, 4, 1, 1, device=99)
    ch = bnr_amd.Chain(X, y, 2, 4, 1, 1)
    ch.init_prior()
    with pytest.raises(bnr_amd.BnrError):
        ch.update("gamma", 1, 2)                                              # row 1 has no predecessor
    with pytest.raises(bnr_amd.BnrError):
        ch.fetch(0, 3)
    ch.close()


def test_headline_size_properties(gpu):
    """n=500, V=100 (q=5050), R=7 (BASELINE.json configs[2]): size-independent properties of the HIP path.
      * determinism: two runs of the same seed are bitwise equal;
      * the carried n-vector bookkeeping (X gamma from the Gram, no third pass over X) agrees with a recomputation
        from scratch: a run split in two calls (second call re-derives the carried sums by an explicit X*gamma pass)
        equals the single-call run to 1e-9;
      * discrete columns take only their legal values, S > 0, pi rows sum to 1, M symmetric positive definite;
      * the first rows equal the oracle's."""
    X, y, _ = bnr_amd.make_synthetic(500, 100, 7, seed=20240501)
    tot = 40
    a = bnr_amd.Chain(X, y, 7, tot, 20240501, 1)
    a.init_prior()
    a.run(2, tot, tot)
    A = a.fetch()
    b = bnr_amd.Chain(X, y, 7, tot, 20240501, 1)
    b.init_prior()
    b.run(2, tot, 17)
    b.run(18, tot, tot)
    B = b.fetch()
    c = bnr_amd.Chain(X, y, 7, tot, 20240501, 1)
    c.init_prior()
    c.run(2, tot, tot)
    Cc = c.fetch()
    for k in bo.COLUMNS:
        assert np.array_equal(A[k], Cc[k]), k
        assert np.allclose(A[k], B[k], rtol=1e-7, atol=1e-10), k
        assert np.all(np.isfinite(A[k])), k
    assert set(np.unique(A["xi"])) <= {0.0, 1.0} and set(np.unique(A["lam"])) <= {0.0, 1.0, -1.0}
    assert np.all(A["S"] > 0) and np.all(A["tau2"] > 0) and np.all(A["theta"] > 0)
    assert np.allclose(A["pi"].sum(axis=2), 1.0)
    for i in range(tot):
        M = A["M"][i]
        assert np.allclose(M, M.T, rtol=1e-10) and np.all(np.linalg.eigvalsh(M) > 0)
        if i > 0:                                      # the init row draws u for every node (gibbs.jl:213-215)
            assert np.all(A["u"][i][:, A["xi"][i, :, 0] == 0] == 0)
    o = bo.Oracle(X, y, 7, 4, 20240501, chain=1, pdf_mode=1)
    o.init_prior()
    o.run(2, 4, 4)
    assert_tables_close(A, o.t, rows_got=slice(0, 4), what="headline first rows")
    # the same chain as member of a lockstep group of five (blockIdx.z path of every kernel at the full size; with five
    # members the trailing update of the factorization runs in its 64 x 64 super-block form, alone in its 32 x 32 form)
    members = [bnr_amd.Chain.like(a, 20240501, cid, tot) for cid in (2, 1, 3, 4, 5)]
    for m in members:
        m.init_prior()
    grp = bnr_amd.Group(members)
    grp.run(2, tot, tot)
    Gm = members[1].fetch()
    for k in bo.COLUMNS:
        assert np.array_equal(Gm[k], A[k]), ("group member vs chain alone", k)
    assert not np.array_equal(members[0].fetch()["gamma"], A["gamma"])
    grp.close()
    for ch in (a, b, c, *members):
        ch.close()


def test_launch_modes_are_equivalent(gpu, test1):
    """Captured hipGraph batches + two-branch schedule (default), eager launches, and the single-stream schedule are
    different ways to issue the same kernels: the traces must be bitwise equal."""
    X, y = test1
    ref = None
    for opts in ({}, {"graph": 0}, {"overlap": 0}, {"graph": 0, "overlap": 0}, {"graph_k": 3}):
        ch = bnr_amd.Chain(X, y, 5, 40, 31, 1)
        for k, v in opts.items():
            ch.set_option(k, v)
        ch.init_prior()
        ch.run(2, 40, 40)
        got = ch.fetch()
        assert ch.counters()["chol_fail"] == 0
        if ref is None:
            ref = got
        else:
            for k in bo.COLUMNS:
                assert np.array_equal(got[k], ref[k]), (opts, k)
        ch.close()


def test_rhat_through_the_library_communicator(gpu, test1):
    """bnr_rhat (return_psrf_VOI, gibbs.jl:771-789, over rhat(), convergence.jl:4-65) with every kind of communicator a one-GPU
    box can hold: none (one process), the library's own RCCL communicator of one rank (ncclCommInitRank + ncclAllGather really
    run), and a host-callback communicator -- all equal the oracle's rhat over the fetched traces."""
    X, y = test1
    chains = [bnr_amd.Chain(X, y, 5, 60, 5, 1)]
    chains += [bnr_amd.Chain.like(chains[0], 5, c, 60) for c in (2, 3)]
    for c in chains:
        c.init_prior()
    g = bnr_amd.Group(chains)
    g.run(2, 20, 60)
    gam = np.stack([c.fetch(21, 60)["gamma"][:, :, 0] for c in chains], axis=2)
    xi = np.stack([c.fetch(21, 60)["xi"][:, :, 0] for c in chains], axis=2)
    want_g, want_x = bo.rhat(gam), bo.rhat(xi)
    from bnr_amd import _capi
    rccl = bnr_amd.Comm.rccl(bnr_amd.Comm.unique_id(), 0, 1, 0)
    assert np.array_equal(rccl.allgather(np.arange(7.0)), np.arange(7.0)[None, :])
    calls = []
    cb = bnr_amd.Comm.callback(0, 1, lambda send: calls.append(send.size) or send[None, :])
    for comm in (None, rccl, cb):
        rg, rx = _capi.rhat(chains, 3, comm, 20, 40)
        assert np.allclose(rg, want_g, rtol=1e-10) and np.allclose(rx, want_x, rtol=1e-10, equal_nan=True)
    with pytest.raises(bnr_amd.BnrError, match="must hold the chains"):
        _capi.rhat(chains[:2], 3, None, 20, 40)
    rccl.close()
    cb.close()
    g.close()
    for c in chains:
        c.close()


def test_gram_kernels_write_the_same_tiles(gpu):
    """k_gram (16-column batches, two workgroups per CU) and k_gram8 (8-column batches, three per CU) are chosen per launch by
    its size; both follow the same K split, K-groups and accumulation order, so a chain's table must not depend on the choice:
    bitwise equal over ragged and tiny shapes (n not a multiple of 64, V = 2, q below one batch, R > V), alone and in a group."""
    shapes = [(40, 8, 3), (1, 2, 1), (3, 2, 4), (65, 5, 2), (129, 12, 7), (200, 50, 5), (64, 33, 10), (500, 100, 7)]
    for n, V, R in shapes:
        X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=77 + n)
        tabs = {}
        for variant in (16, 8):
            ch = bnr_amd.Chain(X, y, R, 6, 3, 1)
            mates = [bnr_amd.Chain.like(ch, 3, c, 6) for c in (2, 3)]
            for c in [ch] + mates:
                c.init_prior()
            g = bnr_amd.Group([mates[0], ch, mates[1]])
            g.set_option("gram_variant", variant)
            g.run(2, 6, 6)
            solo = bnr_amd.Chain.like(ch, 3, 1, 6)
            solo.set_option("gram_variant", variant)
            solo.init_prior()
            solo.run(2, 6, 6)
            tabs[variant] = (ch.fetch(), solo.fetch())
            assert ch.counters()["chol_fail"] == 0
            g.close()
            for c in [ch, solo] + mates:
                c.close()
        for k in bo.COLUMNS:
            assert np.array_equal(tabs[16][0][k], tabs[8][0][k]), ("group", n, V, R, k)
            assert np.array_equal(tabs[16][1][k], tabs[8][1][k]), ("alone", n, V, R, k)
            assert np.array_equal(tabs[16][0][k], tabs[16][1][k]), ("group vs alone", n, V, R, k)


# ------------------------------------------------------------------------------------------------------------------------
# The reference's numerical failure paths, EXECUTED on the device (not only asserted to be absent): crafted rows are loaded with
# bnr_chain_load, one update_*! hook runs on them, and the result, the event counters and the status word are the oracle's.
# Exact arithmetic is used to make the branch taken independent of rounding: powers of two, all-ones loadings.
def _edge_pair(V, R, hyper=None, n=12, seed=5):
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=77)
    ch, o = pair(X, y, R, 3, seed, **(hyper or {}))
    ch.init_prior()
    o.init_prior()
    return ch, o


def _oracle_counts(o):
    return int(o.o.jitter_events), int(o.o.nan_w_events), int(o.o.status)


def test_device_runs_the_jitter_ladder_of_the_node_update(gpu):
    """gibbs.jl:322-347: Sigma^-1 = U'H^-1U/tau2 + inv(M) fails its Cholesky, + 1e-5 on the diagonal passes.  u_v = (1,1), lambda = (1,1),
    S = 1, tau2 = 1, V = 17, M = 2^60 I: Sigma^-1 = 16 [[1,1],[1,1]] + 2^-60 I = 16 [[1,1],[1,1]] exactly (singular: second pivot
    16 - 4^2 = 0), and with the first rung of the ladder the second pivot is 2e-5.  Every node takes the rung once."""
    V, R = 17, 2
    ch, o = _edge_pair(V, R)
    t = o.t
    t["u"][0] = 1.0; t["lam"][0] = 1.0; t["S"][0] = 1.0; t["gamma"][0] = 0.5; t["xi"][0] = 1.0; t["Delta"][0] = 0.5
    t["M"][0] = np.eye(R) * 2.0 ** 60
    t["tau2"][1] = 1.0
    ch.load(t, 1, 2)
    ch.update("u_xi", 2, 2)
    o.update("u_xi", 1, 2)
    jit, nanw, status = _oracle_counts(o)
    c = ch.counters()
    assert jit == V and status == 0 and c["jitter"] == V and c["chol_fail"] == 0 and c["nan_w"] == nanw == 0
    g = ch.fetch(2, 2)
    assert np.array_equal(g["xi"][0], t["xi"][1])
    assert np.allclose(g["u"][0], t["u"][1], rtol=RTOL, atol=ATOL) and np.isfinite(g["u"][0]).all()
    ch.close()


def test_device_reports_a_cholesky_that_fails_after_the_whole_ladder(gpu):
    """The same singular Sigma^-1 at a scale where 1e-5 + 4e-5 vanish in rounding (tau2 = 2^-44: 2^48 [[1,1],[1,1]], square root 2^24 exact, ulp 0.06): both rungs
    fail, the reference rethrows (gibbs.jl:341-346) = status 3 here and in the oracle, with the node update named as the place."""
    V, R = 17, 2
    ch, o = _edge_pair(V, R)
    t = o.t
    t["u"][0] = 1.0; t["lam"][0] = 1.0; t["S"][0] = 1.0; t["gamma"][0] = 0.5; t["xi"][0] = 1.0; t["Delta"][0] = 0.5
    t["M"][0] = np.eye(R) * 2.0 ** 60
    t["tau2"][1] = 2.0 ** -44
    ch.load(t, 1, 2)
    with pytest.raises(bnr_amd.BnrError) as e:
        ch.update("u_xi", 2, 2)
    o.update("u_xi", 1, 2)
    assert e.value.code == 3 == o.status and "node" in str(e.value)
    c = ch.counters()
    assert c["jitter"] >= 1 and c["chol_fail"] >= 1 and c["where"][0] >= 1 and o.o.jitter_events >= 1
    ch.close()


def test_device_flips_the_coin_when_the_node_weight_is_nan(gpu):
    """gibbs.jl:353-360, 392-400: a weight that is NaN -- here through gamma = +Inf on the edge (5,2): c = U'H^-1 gamma is Inf, the two
    substitutions turn it into NaN -- makes xi a fair coin from the XI draw site.  Nodes 2 and 5 take that branch (counted), their xi
    equal the oracle's coin, their u is NaN in both (xi (mu_t + z) with mu_t = NaN); every other node is untouched by it."""
    V, R = 9, 2
    ch, o = _edge_pair(V, R, seed=11)
    t = o.t
    from bnr_amd import _capi
    e52 = _capi.lib().bnr_host_edge_index(V, 5, 2)
    t["gamma"][0, e52, 0] = np.inf
    t["tau2"][1] = 0.7
    ch.load(t, 1, 2)
    ch.update("u_xi", 2, 2)
    o.update("u_xi", 1, 2)
    jit, nanw, status = _oracle_counts(o)
    c = ch.counters()
    assert nanw == 2 == c["nan_w"] and status == 0 and c["chol_fail"] == 0
    g = ch.fetch(2, 2)
    assert np.array_equal(g["xi"][0], t["xi"][1])
    assert np.isnan(g["u"][0][:, [2, 5]]).all() and np.isnan(t["u"][1][:, [2, 5]]).all()
    rest = [v for v in range(V) if v not in (2, 5)]
    assert np.allclose(g["u"][0][:, rest], t["u"][1][:, rest], rtol=RTOL, atol=ATOL) and np.isfinite(g["u"][0][:, rest]).all()
    ch.close()


def test_device_runs_the_retry_of_update_M(gpu):
    """gibbs.jl:529-543: Psi = I + sum u_v u_v' loses its identity in rounding when every u_v = (2^30, 2^30) (V 2^60 >> 2^53): Psi is an exactly
    singular 2 x 2 matrix, the first Cholesky fails, + 1e-5 vanishes as well, the retry fails: jitter counted once, status 3, Psi named."""
    V, R = 16, 2
    ch, o = _edge_pair(V, R)
    t = o.t
    t["u"][1] = 2.0 ** 30
    t["xi"][1] = 1.0
    ch.load(t, 1, 2)
    with pytest.raises(bnr_amd.BnrError) as e:
        ch.update("M", 2, 2)
    o.update("M", 1, 2)
    c = ch.counters()
    assert e.value.code == 3 == o.status and "Psi 1" in str(e.value)
    assert c["jitter"] == 1 == o.o.jitter_events and c["chol_fail"] == 1 and c["where"][1] == 1
    ch.close()


@pytest.mark.parametrize("which", ["chi", "psi"])
def test_device_takes_the_degenerate_gig_branches(gpu, which):
    """gig.jl:15-26: chi < 10 eps (gamma == W on every edge: the draw is Gamma(1/2, scale 2/psi ... as the reference writes it) and
    psi < 10 eps (theta = 1e-300: inverse-Gamma-type branch) -- S of the whole row through the device's update_D! equals the oracle's,
    whose branch counters say which branch every edge took."""
    V, R = 7, 2
    ch, o = _edge_pair(V, R, seed=3)
    t = o.t
    q = V * (V + 1) // 2
    for k in ("u", "lam", "tau2", "gamma", "xi"):
        t[k][1] = t[k][0]                                    # row 2 carries the state update_D! reads (gamma, u of the row; lambda, theta of row 1)
    if which == "chi":
        W = o.compute_W(1, 0)                                # W(u of row 2, lambda of row 1)
        t["gamma"][1, :, 0] = W
    else:
        t["theta"][0] = 1e-300
    ch.load(t, 1, 2)
    before = np.array(o.o.gig_branch[:])
    ch.update("D", 2, 2)
    o.update("D", 1, 2)
    took = np.array(o.o.gig_branch[:]) - before
    assert took.sum() == q and took[3 if which == "chi" else 4] == q, took      # every edge through the degenerate branch in question
    g = ch.fetch(2, 2)
    assert np.isfinite(g["S"][0]).all() and (g["S"][0] > 0).all()
    assert np.allclose(g["S"][0], t["S"][1], rtol=RTOL, atol=0.0)
    c = ch.counters()
    assert c["sampler_cap"] == 0 and c["chol_fail"] == 0
    ch.close()


def test_device_reports_the_sampler_cap(gpu):
    """A rejection sampler that never accepts (Gamma with shape NaN: zeta = NaN in update_theta!) stops at the attempt cap on the device as
    in the oracle: status 4 (BNR_ERR_SAMPLER_CAP), counted."""
    V, R = 5, 2
    ch, o = _edge_pair(V, R, hyper=dict(zeta=float("nan")))
    for k in ("u", "lam", "tau2", "gamma", "xi", "S"):
        o.t[k][1] = o.t[k][0]
    ch.load(o.t, 1, 2)
    with pytest.raises(bnr_amd.BnrError) as e:
        ch.update("theta", 2, 2)
    o.update("theta", 1, 2)
    assert e.value.code == 4 == o.status
    assert ch.counters()["sampler_cap"] >= 1
    # ... reported by the call in which it happened and only by that one: the device counter is cumulative, the rows were written with the
    # samplers' fall-backs and the table stays usable (the reference's rejection loops are unbounded and never raise here) -- the next
    # call on the chain succeeds, the counter keeps the total
    ch.update("Delta", 2, 2)
    assert ch.counters()["sampler_cap"] >= 1
    ch.close()


def test_group_reports_the_most_severe_member_status(gpu):
    """bnr_group_run must not let a member's sampler cap (status 4, a warning to the callers: gibbs.jl's rejection loops never raise) hide another member's failed
    factorization (status 3; the reference's `\\` of gibbs.jl:434 would throw) in the same call.  Member 1: zeta = NaN, update_theta!'s Gamma never accepts -> cap.  Member 2: an
    infinite S in its first row -> G + I is not finite -> the factorization reports it.  Whichever order the members are in, the call returns 3."""
    n, V, R = 40, 6, 2
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=3)
    for order in (0, 1):
        capped = bnr_amd.Chain(X, y, R, 4, 11, 1, zeta=float("nan"))
        broken = bnr_amd.Chain(X, y, R, 4, 11, 2)
        for c in (capped, broken):
            c.init_prior()
        t = broken.fetch(1, 1)
        t["S"][0, 3] = np.inf
        broken.load(t, 1, 1)
        g = bnr_amd.Group([capped, broken] if order == 0 else [broken, capped])
        with pytest.raises(bnr_amd.BnrError) as e:
            g.run(2, 2, 2)
        assert e.value.code == 3 and "Cholesky" in str(e.value), (order, e.value.code, str(e.value))
        assert capped.counters()["sampler_cap"] >= 1 and broken.counters()["chol_fail"] >= 1
        g.close()
        capped.close(); broken.close()


def test_a_non_finite_S_is_reported_on_the_i8_gram_path_too(gpu):
    """the fixed-point image of S that feeds the i8 Gram of a binary model matrix (k_sdigits) cannot carry a NaN or an infinity -- fmax drops the one, the conversion of
    the other is undefined -- where the f64 Gram hands either to the factorization, which reports it (status 3, G + I).  Both paths must end the call the same way."""
    n, V, R = 70, 12, 3
    rng = np.random.default_rng(8)
    X = bnr_amd.XInput(np.asfortranarray(rng.random((n, V * (V + 1) // 2)) < 0.5), False)
    y = rng.normal(size=n)
    for gram_i8 in (1, 0):
        for bad in (np.nan, np.inf):
            ch = bnr_amd.Chain(X, y, R, 4, 5, 1)
            ch.set_option("gram_i8", gram_i8)
            ch.init_prior()
            t = ch.fetch(1, 1)
            t["S"][0, 7] = bad
            ch.load(t, 1, 1)
            with pytest.raises(bnr_amd.BnrError) as e:
                ch.run(2, 2, 2)
            assert e.value.code == 3, (gram_i8, bad, e.value.code, str(e.value))
            assert ch.counters()["chol_fail"] >= 1
            ch.close()


def test_headline_size_posterior_summaries_match_the_cpu_restatement(gpu):
    """Long-horizon pin at the HEADLINE size (until round 6 the only such check ran at n = 70): BASELINE configs[2]'s fit -- n=500, V=100, R=7, 8 chains,
    seed 20240501 -- 1000 burn-in + 2000 kept sweeps per chain on the GPU as one lockstep group, against the same fit run by the CPU restatement of the reference
    (tests/golden/headline_pin.npz, written by tests/golden/make_headline_pin.py; gibbs.jl:191-677).  Same seeds, same draw sites: the two sample paths coincide until
    rounding flips a discrete draw somewhere, after which they are two runs of the same sampler -- so per chain P(xi_v = 1), the means of gamma on a fixed subset of 389
    edges and the mean of tau2 must agree within 6 Monte-Carlo standard errors of their difference (batch means, 20 batches of 100 sweeps), and on average much closer."""
    f = os.path.join(G, "headline_pin.npz")
    pin = np.load(f)
    n, V, R, seed, nburn, nsamp = (int(pin[k]) for k in ("n", "V", "R", "seed", "nburn", "nsamp"))
    edges = pin["edges"]
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=seed)
    tot = nburn + nsamp + 1
    chains = [bnr_amd.Chain(X, y, R, tot, seed, 1)]
    chains += [bnr_amd.Chain.like(chains[0], seed, c, tot) for c in range(2, 9)]
    for ch in chains:
        ch.init_prior()
    g = bnr_amd.Group(chains)
    g.run(2, tot, tot)
    nb = pin["xi_batch"].shape[1]
    bs = nsamp // nb
    zs, close = [], 0
    for c, ch in enumerate(chains):
        t = ch.fetch(nburn + 2, tot)
        assert ch.counters()["chol_fail"] == 0
        xi = t["xi"][:, :, 0]
        ga = t["gamma"][:, :, 0][:, edges]
        t2 = t["tau2"][:, 0, 0]
        for name, mine, ref_mean, ref_batch in (("xi", xi, pin["xi_mean"][c], pin["xi_batch"][c]), ("gamma", ga, pin["gamma_mean"][c], pin["gamma_batch"][c]),
                                                ("tau2", t2[:, None], pin["tau2_mean"][c:c + 1], pin["tau2_batch"][c][:, None])):
            m = mine.mean(0)
            mb = mine.reshape(nb, bs, -1).mean(1)
            se = np.sqrt((mb.var(0, ddof=1) + ref_batch.var(0, ddof=1)) / nb) + 1e-12 + 1e-9 * np.abs(ref_mean)
            z = np.abs(m - ref_mean) / se
            assert np.all(z < 6.0), (c + 1, name, float(z.max()), int(z.argmax()))
            zs.append(z)
            close += int(np.sum(np.abs(m - ref_mean) <= 1e-6 * (1e-3 + np.abs(ref_mean))))
    z = np.concatenate(zs)
    assert np.mean(z) < 1.5, float(np.mean(z))                # (two independent runs would give E|z| = 0.8)
    print("headline pin: %d summaries over 8 chains, max z %.2f, mean z %.2f, %d of them equal to 1e-6 (paths that never parted)" % (z.size, z.max(), z.mean(), close))
    g.close()
    for ch in chains:
        ch.close()


def test_post_burn_in_rows_match_the_oracle_at_config3(gpu):
    """Full-size parity AFTER burn-in (most xi = 0, S small, the Gram well conditioned -- another regime than the rows right after the
    prior draw): the 8 chains of BASELINE configs[2] run 2 000 sweeps as one lockstep group on the GPU, then the oracle continues from
    the last row of chain 1 (same table, same iteration counter) for 3 rows, and the GPU's next 3 rows equal them (RTOL 1e-6)."""
    n, V, R, tot = 500, 100, 7, 2004
    X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=20240501)
    chains = [bnr_amd.Chain(X, y, R, tot, 20240501, 1)]
    chains += [bnr_amd.Chain.like(chains[0], 20240501, c, tot) for c in range(2, 9)]
    for c in chains:
        c.init_prior()
    g = bnr_amd.Group(chains)
    g.run(2, tot, 2001)
    last = chains[0].fetch(2001, 2001)
    assert (np.abs(last["xi"][0]) < 0.5).mean() > 0.3                # a post-burn-in state: a good share of the nodes switched off
    table = bo.new_table(4, V, R)
    for k in bo.COLUMNS:
        table[k][0] = last[k][0]
    o = bo.Oracle(X, y, R, 4, 20240501, chain=1, pdf_mode=1, table=table)
    o.iter = 2001
    g.run(2002, tot, tot)
    o.run(2, 4, 4)
    got = chains[0].fetch(2001, 2004)
    assert_tables_close(got, o.t, what="config 3, rows 2002-2004 of chain 1")
    cnt = chains[0].counters()
    assert cnt["chol_fail"] == 0 and cnt["sampler_cap"] == 0
    g.close()
    for c in chains:
        c.close()


def test_experimental_options_are_refused_by_the_shipped_library(gpu, test1):
    """The measured experiments of rounds 3 and 4 (left-looking factorization behind progress gates, persistent / resident Gram kernels with task
    queues and reserved CUs, the pipelined and "linear" schedules, the group back-projection) were removed from the tree in round 5
    (tools/experiments/ keeps their kernels and drivers for the record); part of them polled device memory.  The library has none of them:
    every option that would select one is refused by name with a pointer to that directory, for a chain and for a group, and the chain works
    on as before."""
    X, y = test1
    ch = bnr_amd.Chain(X, y, 5, 8, 3, 1)
    mate = bnr_amd.Chain.like(ch, 3, 2, 8)
    g = bnr_amd.Group([ch, mate])
    for target in (ch, g):
        for name, value in (("pipeline", 1), ("linear", 2), ("linear_merge", 1), ("gate_us", 100), ("group_backproj", 1), ("resv_mask", 0x80),
                            ("crit_origin", 1), ("gram_variant", 9), ("gram_variant", 10), ("gram_variant", 13), ("factor_variant", 1), ("factor_variant", 4), ("factor_variant", 5), ("nop_fork", 1)):
            with pytest.raises(bnr_amd.BnrError, match="tools/experiments"):
                target.set_option(name, value)
    assert bnr_amd.lib().bnr_debug_set_exp(0, 1) != 0
    for c in (ch, mate):
        c.init_prior()
    g.run(2, 8, 8)
    assert ch.counters()["chol_fail"] == 0 and np.isfinite(ch.fetch()["gamma"]).all()
    g.close()
    ch.close(); mate.close()


def test_group_of_one_survives_a_resize(gpu, test1):
    """A lockstep group with ONE member issues its kernels with the member's descriptor by value, baked into the group's captured graphs
    (bnr_one): resizing the member's table (generate_samples_dbl! re-allocates it every round, gibbs.jl:1164-1172) must drop those graphs too
    (bnr_hip.hip, bnr_chain_resize / ensure_plan).  Group of one: run, resize, run == the same chain run alone, bitwise."""
    X, y = test1
    solo = bnr_amd.Chain(X, y, 5, 40, 17, 1)
    solo.init_prior()
    solo.run(2, 20, 20)
    solo.resize(40 + 25)
    solo.run(21, 65, 65)
    want = solo.fetch()
    ch = bnr_amd.Chain.like(solo, 17, 1, 40)
    ch.init_prior()
    g = bnr_amd.Group([ch])
    g.prepare()                           # (round 4: a group of ONE prepared before its first run replayed its graphs on an unfilled descriptor array)
    g.run(2, 20, 20)                      # replays the group's graphs (descriptor of the 40-row table inside)
    ch.resize(40 + 25)                    # new trace allocation: stale graphs would write into freed memory
    g.run(21, 65, 65)                     # (no prepare in between: a prepare re-derives the carried sums, equal only to rounding)
    got = ch.fetch()
    for k in bo.COLUMNS:
        assert np.array_equal(got[k], want[k]), k
    # a plan longer than the chain's first allocation regrows it (ensure_plan): the group's graphs are dropped there as well
    ch.resize(70000)
    g.prepare()
    g.run(66, 69999, 69999)
    assert ch.counters()["chol_fail"] == 0 and np.isfinite(ch.fetch(69999, 69999)["gamma"]).all()
    g.close()
    ch.close(); solo.close()


def test_factorization_variants_are_bitwise_equal(gpu):
    """k_chol_step2 (two panels per launch; variant 3: the trailing matrix updated every other launch with K = 128) gives bitwise the tables of the
    default right-looking path (gibbs.jl:434), alone and as members of a lockstep group, from graphs and eagerly; so does the default's first
    launch, which sums the Gram's K-split partial tiles itself (fuse_reduce), against the separate k_gram_reduce pass, and the group's X pass
    with one workgroup per column chunk for all members (group_xpass) against the per-chain kernel, and the partial sums as a launch of their
    own (split_sums).  (The experimental variants -- left-looking factorization, resident Gram kernels, pipelined schedule, group
    back-projection -- were checked the same way by tools/experiments/ab_factor.py against the round-4 experiments build.)"""
    for (n, V, R) in [(70, 19, 5), (193, 30, 5), (64, 9, 2), (500, 40, 4), (1000, 12, 3)]:
        X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=7)
        tabs = {}
        for name, opts in (("right", {"factor_variant": 0}), ("right, separate reduction pass", {"factor_variant": 0, "fuse_reduce": 0}),
                           ("right, separate reduction pass, eager", {"factor_variant": 0, "fuse_reduce": 0, "graph": 0}), ("right, eager", {"factor_variant": 0, "graph": 0}),
                           ("right, X pass per chain", {"factor_variant": 0, "group_xpass": 0}),
                           ("right, X pass for the group", {"factor_variant": 0, "group_xpass": 1}),
                           ("right, sums split off", {"factor_variant": 0, "split_sums": 1}), ("right, sums inside", {"factor_variant": 0, "split_sums": 0}),
                           ("two panels", {"factor_variant": 2}), ("two panels, eager", {"factor_variant": 2, "graph": 0}),
                           ("two panels, K = 128 trailing update", {"factor_variant": 3}), ("two panels, K = 128, eager", {"factor_variant": 3, "graph": 0})):
            ch = bnr_amd.Chain(X, y, R, 6, 3, 1)
            mates = [bnr_amd.Chain.like(ch, 3, c, 6) for c in (2, 3)]
            for c in [ch] + mates:
                c.init_prior()
            g = bnr_amd.Group([mates[0], ch, mates[1]])
            solo = bnr_amd.Chain.like(ch, 3, 1, 6)
            solo.init_prior()
            for k, v in opts.items():
                g.set_option(k, v)
                solo.set_option(k, v)
            g.run(2, 6, 6)
            solo.run(2, 6, 6)
            tabs[name] = (ch.fetch(), solo.fetch())
            assert ch.counters()["chol_fail"] == 0 and solo.counters()["chol_fail"] == 0
            g.close()
            for c in [ch, solo] + mates:
                c.close()
        # Round 6: the one-panel kernel's sweep is a pipeline of the workgroup's four waves (bnr_panel_sweep_pipe: rank-1 updates column by column), the two-panel kernel
        # keeps the single sweeping wave with an MFMA update between its halves -- two summation orders of the same factor.  Bitwise equality holds inside each family
        # (every schedule / launch mode / group membership of a kernel); across the families the tables agree to rounding.
        for name, (grp, alone) in tabs.items():
            ref = tabs["two panels"] if name.startswith("two panels") else tabs["right"]
            for k in bo.COLUMNS:
                assert np.array_equal(grp[k], ref[0][k]), (name, "group", n, V, R, k)
                assert np.array_equal(alone[k], ref[1][k]), (name, "alone", n, V, R, k)
                assert np.array_equal(grp[k], alone[k]), (name, "group vs alone", n, V, R, k)
        for k in bo.COLUMNS:
            a, b = tabs["two panels"][1][k], tabs["right"][1][k]
            if k in ("xi", "lam"):
                assert np.array_equal(a, b), ("families", n, V, R, k)
            else:
                assert np.allclose(a, b, rtol=1e-7, atol=1e-9), ("families", n, V, R, k, float(np.max(np.abs(a - b))))


def test_back_projection_with_one_edge_per_lane_is_bitwise_equal(gpu):
    """k_backproj64 (a workgroup owns 64 edges, its drawing wave one edge per lane and the reference's own attempt loop: the default for launches of many rounds of
    workgroups -- a lockstep group at large q) writes bitwise the tables of k_backproj (gibbs.jl:435-436, 454-458, 603-605): gamma, S and the per-chunk partial sums, alone
    and as members of a group, from graphs and eagerly, with the sums inside the launch and as a launch of their own, on the f64 and on the byte image of X; edge counts that
    end in a short chunk, in a workgroup with one chunk only, and a single chunk.  One more sweep from a state with theta = 1e-300 sends every edge through the degenerate
    branch (gig.jl:21-26) in both kernels."""
    for (n, V, R, binary) in [(70, 19, 5, False), (193, 30, 5, False), (64, 7, 2, False), (500, 40, 4, True), (130, 12, 3, False), (64, 150, 3, False), (40, 23, 11, False)]:
        if binary:
            rng = np.random.default_rng(5)
            X = bnr_amd.XInput(np.asfortranarray(rng.random((n, V * (V + 1) // 2)) < 0.5), False)
            y = rng.normal(size=n)
        else:
            X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=11)
        tabs = {}
        hyper = dict(nu=R + 2) if R > 8 else {}
        for name, opts in (("32 edges per workgroup", {"wide_backproj": 0}), ("64 edges", {"wide_backproj": 1}), ("64 edges, eager", {"wide_backproj": 1, "graph": 0}),
                           ("64 edges, sums split off", {"wide_backproj": 1, "split_sums": 1}), ("64 edges, sums inside", {"wide_backproj": 1, "split_sums": 0})):
            ch = bnr_amd.Chain(X, y, R, 7, 3, 1, **hyper)
            mates = [bnr_amd.Chain.like(ch, 3, c, 7) for c in (2, 3)]
            solo = bnr_amd.Chain.like(ch, 3, 1, 7)
            for c in [ch, solo] + mates:
                c.init_prior()
                if binary:
                    c.set_option("gram_i8", 0)
            g = bnr_amd.Group([mates[0], ch, mates[1]])
            for k, v in opts.items():
                g.set_option(k, v)
                solo.set_option(k, v)
            g.run(2, 6, 6)
            solo.run(2, 6, 6)
            t = solo.fetch()
            t["theta"][5] = 1e-300                             # row 6: psi < 10 eps in the next sweep's update_D!
            solo.load(t, 6, 6)
            solo.run(7, 7, 7)
            tabs[name] = (ch.fetch(1, 6), solo.fetch())
            assert ch.counters()["chol_fail"] == 0 and solo.counters()["chol_fail"] == 0 and solo.counters()["sampler_cap"] == 0
            g.close()
            for c in [ch, solo] + mates:
                c.close()
        base = tabs["32 edges per workgroup"]
        assert np.isfinite(base[1]["S"]).all() and (base[1]["S"][6] > 0).all()
        for name, (grp, alone) in tabs.items():
            for k in bo.COLUMNS:
                assert np.array_equal(grp[k], base[0][k]), (name, "group", n, V, R, k)
                assert np.array_equal(alone[k], base[1][k]), (name, "alone", n, V, R, k)
                assert np.array_equal(grp[k][:5], alone[k][:5]), (name, "group vs alone", n, V, R, k)


def test_prepare_never_changes_results(gpu, test1):
    """bnr_chain_prepare / bnr_group_prepare capture the graphs and replay them once on scratch rows: tables, iteration
    counters and event counters of a chain alone and of a lockstep group are bitwise what they are without it -- called
    before init_prior (nothing to replay from), after it, and between two run calls."""
    X, y = test1
    ref = bnr_amd.Chain(X, y, 5, 40, 31, 1)
    ref.init_prior()
    ref.run(2, 40, 40)
    want = ref.fetch()
    a = bnr_amd.Chain.like(ref, 31, 1, 40)
    a.prepare()
    a.init_prior()
    a.prepare()
    a.run(2, 40, 17)
    a.prepare()
    a.prepare()
    a.run(18, 40, 40)
    members = [bnr_amd.Chain.like(ref, 31, c, 40) for c in (2, 1, 3)]
    for m in members:
        m.init_prior()
    g = bnr_amd.Group(members)
    g.prepare()
    g.run(2, 40, 9)
    g.prepare()
    g.run(10, 40, 40)
    for got, what in ((a.fetch(), "chain"), (members[1].fetch(), "group member")):
        for k in bo.COLUMNS:
            assert np.allclose(got[k], want[k], rtol=1e-9, atol=1e-12), (what, k)   # split runs re-derive the carried sums
    b = bnr_amd.Chain.like(ref, 31, 1, 40)
    b.init_prior()
    b.prepare()
    b.run(2, 40, 40)
    for k in bo.COLUMNS:
        assert np.array_equal(b.fetch()[k], want[k]), k
    assert a.iter == ref.iter == b.iter and a.counters() == ref.counters()
    g.close()
    for ch in [ref, a, b] + members:
        ch.close()


def test_progress_callback_ticks_like_run(gpu, test1):
    """Chain 1 ticks every prog_freq iterations (gibbs.jl:854-856); the trace does not depend on the tick frequency."""
    X, y = test1
    ticks = []
    a = bnr_amd.Chain(X, y, 5, 50, 8, 1)
    a.init_prior()
    a.run(2, 50, 50, prog_freq=10, callback=lambda done: ticks.append(done))
    assert ticks == [9, 19, 29, 39, 49]          # i = 10, 20, ... completed -> iterations done so far in this call
    b = bnr_amd.Chain(X, y, 5, 50, 8, 1)
    b.init_prior()
    b.run(2, 50, 50)
    A, B = a.fetch(), b.fetch()
    for k in bo.COLUMNS:
        assert np.array_equal(A[k], B[k]), k
    a.close()
    b.close()
