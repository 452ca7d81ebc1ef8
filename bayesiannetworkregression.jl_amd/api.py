"""Host-side mirror of the reference's public and chain-driver API for the Gibbs hot path, over the C ABI.

SPDX-License-Identifier: GPL-2.0-or-later.  Provenance: `generate_samples` / `generate_samples_dbl` restate the schedule arithmetic of
BayesianNetworkRegression.jl (GPL-2.0; S. Ozminkowski, C. Solis-Lemus), src/gibbs.jl:955-1013 and 1119-1190 -- the `num2move` rules,
`tot_sze`, `first_index` and the messages -- because a drop-in for that API must walk the same rows; the rest is written for this repository.

Julia is not available in this image, so the thin host layer a Julia user would get from julia/BNRHip.jl
(ccall) is mirrored here in Python (ctypes) with the reference's names, argument meaning and defaults:

  Fit!                   gibbs.jl:725-751   -> Fit
  generate_samples!      gibbs.jl:897-1020  -> generate_samples
  generate_samples_dbl!  gibbs.jl:1051-1198 -> generate_samples_dbl
  initialize_and_run!    gibbs.jl:822-846   -> initialize_and_run
  run!                   gibbs.jl:849-864   -> run
  return_psrf_VOI        gibbs.jl:771-789   -> return_psrf_VOI
  Results / BNRSummary   gibbs.jl:23-43     -> Results / BNRSummary
  Summary                gibbs.jl:1214-1250 -> Summary
  lower_triangle / create_lower_tri / setup_X!  utils.jl:17-57, gibbs.jl:239-247

All sampling runs on the GPU through libbnr_hip.so; this file holds only the schedule logic (chain fan-out,
PSRF-driven top-ups), the table layout and the post-processing that the reference also does on the host.
Chains are placed one (or several) per GPU; with torch.distributed initialised, chains are sharded over ranks and
the per-chain split-Rhat statistics are all-gathered (RCCL on GPUs, gloo on CPU tests).
"""
import datetime
import math
import random
import sys
from dataclasses import dataclass

import numpy as np

from . import _capi
from ._capi import Chain, Comm, Group, XInput, ess_from_stats, new_table, rhat_from_stats

CITATION = ("If you use BayesianNetworkRegression.jl, please cite:\n@article{Ozminkowski2022,\n"
            "author = {Ozminkowski, S. and Sol\\'{i}s-Lemus, C.},\nyear = {2022},\n"
            "title = {{Identifying microbial drivers in biological phenotypes with a Bayesian Network Regression model}},\n"
            "journal = {In preparation}\n}")


# ------------------------------------------------------------------------------------------ utils.jl
def lower_triangle(matrix):
    """utils.jl:40-57: column-wise lower triangle INCLUDING the diagonal; reads matrix[j,i], j>=i."""
    m = np.asarray(matrix)
    if m.shape[0] != m.shape[1]:
        raise ValueError("matrix must be square")
    V = m.shape[0]
    return np.concatenate([m[i:, i] for i in range(V)])


def create_lower_tri(vector, V):
    """utils.jl:17-27: inverse of lower_triangle (upper part zero)."""
    v = np.asarray(vector).reshape(-1)
    mat = np.zeros((V, V), dtype=v.dtype)
    i = 0
    for k in range(V):
        mat[k:, k] = v[i:i + V - k]
        i += V - k
    return mat


def setup_X(X, x_transform=True):
    """setup_X! (gibbs.jl:239-247) + the V,q bookkeeping of generate_samples! (907-918) -> (X_new n x q float64, V, q)."""
    if x_transform:
        V = np.asarray(X[0]).shape[0]
        q = V * (V + 1) // 2
        X_new = np.empty((len(X), q), dtype=np.float64, order="F")
        for i in range(len(X)):
            X_new[i, :] = lower_triangle(X[i])
    else:
        X_new = np.asfortranarray(X, dtype=np.float64)
        q = X_new.shape[1]
        V = int((-1 + math.sqrt(1 + 8 * q)) / 2)
    return X_new, V, q


# ------------------------------------------------------------------------------------------ output structs
@dataclass
class Results:
    """gibbs.jl:23-29.  state: dict of arrays (tot_save,d1,d2) in the reference layout (chain 1 only)."""
    state: dict
    rhatxi: np.ndarray
    rhatgamma: np.ndarray
    burn_in: int
    sampled: int
    summary_device: dict = None      # filled on request: Summary statistics computed on the GPU (see Summary)
    essxi: np.ndarray = None         # filled on request (ess_max_lag=...): bulk effective sample sizes over all chains
    essgamma: np.ndarray = None


@dataclass
class BNRSummary:
    """gibbs.jl:39-43 (DataFrames replaced by dicts of columns)."""
    edge_coef: dict
    prob_nodes: dict
    ci_level: int

    def __str__(self):
        e, p = self.edge_coef, self.prob_nodes
        lines = ["", "Edge Coefficient Estimates (%d%% credible intervals)" % self.ci_level,
                 " node1 node2 estimate lower_bound upper_bound"]
        for i in range(len(e["node1"])):
            lines.append(" %5d %5d %8.3f %11.3f %11.3f" % (e["node1"][i], e["node2"][i], e["estimate"][i], e["lower_bound"][i], e["upper_bound"][i]))
        lines.append("Node Probabilities")
        lines += [" %.3f" % v for v in p["probability"]]
        return "\n".join(lines)


def _julia_round(x):
    """Julia's round(): half to even, like Python's round() on floats."""
    return int(round(x))


def _summary_ranks(nsamp, interval):
    """1-based positions in the sorted sample the reference reads (gibbs.jl:1224-1233)."""
    lower_bound = (100 - interval) / 200
    upper_bound = 1 - lower_bound
    lw, hi = _julia_round(nsamp * lower_bound), _julia_round(nsamp * upper_bound)
    if lw < 1 or hi > nsamp:
        # the reference indexes the sorted vector at 0 here and stops with a BoundsError
        raise IndexError("Summary: %d samples are too few for a %s%% interval" % (nsamp, interval))
    return lw, hi


def device_summary(chain, nburn, nsamp, interval=95):
    """Summary statistics of one chain's table computed on the GPU (bnr_chain_summary): only 3q + V numbers cross PCIe."""
    lw, hi = _summary_ranks(nsamp, interval)
    mean, lo, up, pxi = chain.summary(nburn + 1, nsamp, lw, hi)
    return dict(interval=interval, estimate=mean, lower_bound=lo, upper_bound=up, probability=pxi)


def Summary(results, interval=95, digits=3):
    """gibbs.jl:1214-1250.  Uses the statistics computed on the GPU when the fit carried them (summary_interval=...),
    otherwise sorts the fetched gamma trace on the host like the reference."""
    nburn, nsamp = results.burn_in, results.sampled
    total = nburn + nsamp
    dev = results.summary_device
    if dev is not None and dev["interval"] == interval:
        est, lo, up, pxi = dev["estimate"], dev["lower_bound"], dev["upper_bound"], dev["probability"]
    else:
        g = results.state["gamma"][nburn:total, :, 0]
        g_sorted = np.sort(g, axis=0)
        lw, hi = _summary_ranks(nsamp, interval)
        est, lo, up = g.mean(axis=0), g_sorted[lw - 1, :], g_sorted[hi - 1, :]
        pxi = results.state["xi"][nburn:total, :, 0].mean(axis=0)
    q = est.shape[0]
    V = int((-1 + math.sqrt(1 + 8 * q)) / 2)
    node1, node2 = [], []
    for k in range(1, V + 1):
        for l in range(k, V + 1):
            node1.append(k)
            node2.append(l)
    edge = dict(node1=np.array(node1), node2=np.array(node2), estimate=np.round(est, digits),
                lower_bound=np.round(lo, digits), upper_bound=np.round(up, digits))
    xi = dict(probability=np.round(pxi, digits))
    return BNRSummary(edge, xi, interval)


# ------------------------------------------------------------------------------------------ chain placement
def _dist():
    """torch.distributed if THE CALLER has imported and initialised it, else None.  Never imports torch itself: a process group can only
    exist if the caller imported torch already, and importing it here would map the torch wheel's own libamdhip64 in front of
    libbnr_hip.so's (/opt/rocm) in a process that has not created a chain yet -- the load-order guard of _capi.lib() then refuses to run."""
    import sys
    dist = sys.modules.get("torch.distributed")
    try:
        if dist is not None and dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        pass
    return None


def _rank_world():
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


def local_chain_ids(num_chains):
    """Chains c = 1..num_chains are sharded round-robin over ranks (the reference's pmap, gibbs.jl:946)."""
    rank, world = _rank_world()
    return [c for c in range(1, num_chains + 1) if (c - 1) % world == rank]


def _default_device():
    import os
    return int(os.environ.get("LOCAL_RANK", "0")) if _dist() else 0


def shared_seed(seed, draw):
    """The ONE seed of a fit: the reference draws it once and keys chain c with seed + c (gibbs.jl:739, 928).  With chains
    sharded over torch.distributed ranks the draw happens on rank 0 and is broadcast -- independent draws per rank could put
    two chains on the same stream (s_a + c_a == s_b + c_b) and would make parameters.log's seed describe rank 0 only."""
    d = _dist()
    if seed is not None and d is None:
        return int(seed)
    value = int(draw()) if seed is None else int(seed)
    if d is None:
        return value
    import torch
    use_cuda = d.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.int64, device=dev)
    d.broadcast(t, src=0)
    return int(t.item())


def make_comm(device=None, force=False):
    """The communicator of this fit for the library's own exchange (bnr_rhat): None without torch.distributed; with backend
    nccl an RCCL communicator OWNED BY THE LIBRARY (rank 0's unique id travels through torch.distributed's object broadcast
    -- the only thing torch is used for here); otherwise (gloo: CPU tests, one-GPU rehearsals) a callback communicator whose
    all-gather runs over torch.distributed."""
    d = _dist()
    if d is None or (d.get_world_size() == 1 and not force):
        return None
    import torch
    rank, world = d.get_rank(), d.get_world_size()
    if d.get_backend() == "nccl":
        box = [Comm.unique_id() if rank == 0 else None]
        d.broadcast_object_list(box, src=0)
        return Comm.rccl(box[0], rank, world, torch.cuda.current_device() if device is None else device)

    def gather(send):
        t = torch.from_numpy(send)
        out = [torch.empty_like(t) for _ in range(world)]
        d.all_gather(out, t)
        return torch.stack(out).numpy()
    return Comm.callback(rank, world, gather)


def allgather_stats(local_stats, num_chains, comm=None):
    """All-gather of per-chain messages of equal width (the ESS message; 4*(q+V) doubles for split-Rhat).  local_stats:
    {chain_id: array}.  Returns (num_chains, width) in chain order on every rank.  With a library communicator (make_comm)
    the exchange is bnr_comm_allgather (RCCL owned by the library, or the host callback); without one it falls back to
    torch.distributed's all_gather (RCCL on GPUs, gloo on CPU tests)."""
    d = _dist()
    width = len(next(iter(local_stats.values()))) if local_stats else 0
    if comm is not None:
        world = comm.world
        per_rank = (num_chains + world - 1) // world
        width = int(comm.allgather(np.array([float(width)])).max())
        buf = np.zeros((per_rank, width))
        for slot, c in enumerate(sorted(local_stats)):
            buf[slot] = local_stats[c]
        allv = comm.allgather(buf.reshape(-1)).reshape(world, per_rank, width)
        return np.stack([allv[(c - 1) % world, (c - 1) // world] for c in range(1, num_chains + 1)])
    if d is None:
        return np.stack([local_stats[c] for c in range(1, num_chains + 1)])
    import torch
    rank, world = _rank_world()
    per_rank = (num_chains + world - 1) // world
    wt = torch.tensor([width], dtype=torch.int64)
    use_cuda = d.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    wt = wt.to(dev)
    d.all_reduce(wt, op=d.ReduceOp.MAX)
    width = int(wt.item())
    buf = torch.zeros(per_rank, width, dtype=torch.float64)
    for slot, c in enumerate(sorted(local_stats)):
        buf[slot] = torch.from_numpy(np.asarray(local_stats[c], dtype=np.float64))
    buf = buf.to(dev)
    out = [torch.empty_like(buf) for _ in range(world)]
    d.all_gather(out, buf)
    allv = torch.stack(out).cpu().numpy()          # (world, per_rank, width)
    res = np.empty((num_chains, width))
    for c in range(1, num_chains + 1):
        res[c - 1] = allv[(c - 1) % world, (c - 1) // world]
    return res


class ChainSet:
    """The chains of one fit that live on this rank's GPU."""

    def __init__(self, X_new, y, R, num_chains, tot_save, seed, hyper, device=None):
        self.num_chains = num_chains
        self.ids = local_chain_ids(num_chains)
        dev = _default_device() if device is None else device
        self.chains = {}
        for c in self.ids:                                  # X, y are uploaded once per GPU and shared by its chains
            first = next(iter(self.chains.values()), None)
            self.chains[c] = Chain(X_new, y, R, tot_save, seed, c, device=dev, **hyper) if first is None else Chain.like(first, seed, c, tot_save)
        # several chains share this GPU: they advance in lockstep, one launch per kernel for all of them (the sequential
        # panel chain of the n x n factorization is paid once per sweep of the whole group)
        self.group = Group([self.chains[c] for c in self.ids]) if len(self.chains) > 1 else None
        self.V, self.q, self.R = (next(iter(self.chains.values())).V, next(iter(self.chains.values())).q, R) if self.chains else (None, None, R)
        if self.V is None:                                  # a rank without a chain still takes part in the exchanges
            self.q = X_new.q if isinstance(X_new, XInput) else int(np.asarray(X_new).shape[1])
            self.V = int((-1 + math.sqrt(1 + 8 * self.q)) / 2)
        self.comm = make_comm(dev)

    def init_prior(self):
        for ch in self.chains.values():
            ch.init_prior()

    def run(self, first_index, nburn, total, purge_burn, prog_freq=0, callback=None):
        """run! on every local chain (in lockstep when there are several).  The rank that holds chain 1 ticks the
        progress callback (gibbs.jl:854-856)."""
        cb = callback if 1 in self.ids else None

        def tolerant(call):
            # BNR_ERR_SAMPLER_CAP (4): a rejection sampler stopped at its attempt cap somewhere in this call.  Every row was written (with the
            # capped draw's last proposal) and the table stays valid; the reference has no cap and never raises here.  A fit of 50 000
            # iterations must not die of one such draw: warn once per call and go on -- counters()['sampler_cap'] keeps the total.
            try:
                call()
            except _capi.BnrError as e:
                if getattr(e, "code", None) == _capi.BNR_ERR_SAMPLER_CAP:
                    import warnings
                    warnings.warn("libbnr_hip: a rejection sampler hit its attempt cap during rows %d..%d (the rows were written; see counters()['sampler_cap'])" % (first_index, total), RuntimeWarning)
                else:
                    raise

        if self.group is not None:
            tolerant(lambda: self.group.run(first_index, nburn, total, purge_burn, prog_freq, cb))
        else:
            for c in self.ids:
                tolerant(lambda c=c: self.chains[c].run(first_index, nburn, total, purge_burn, prog_freq, cb if c == 1 else None))

    def rhat(self, first_row, nsamp):
        """split-Rhat over ALL chains of the fit for gamma (q) then xi (V) (return_psrf_VOI, gibbs.jl:771-789)."""
        return _capi.rhat([self.chains[c] for c in self.ids], self.num_chains, self.comm, first_row - 1, nsamp, self.V, self.q)

    def ess(self, first_row, nsamp, max_lag=None):
        """Bulk effective sample size over ALL chains of the fit for gamma (q) then xi (V) -- an addition to the reference
        (split chains + Geyer's sequence, as Stan / MCMCDiagnosticTools.ess); same exchange pattern as rhat()."""
        max_lag = min(250, nsamp // 4) if max_lag is None else max_lag
        local = {c: ch.ess_stats(first_row, nsamp, max_lag) for c, ch in self.chains.items()}
        allst = allgather_stats(local, self.num_chains, self.comm)
        e = ess_from_stats(allst, nsamp, max_lag)
        return e[:self.q], e[self.q:]

    @staticmethod
    def _V_from_width(npar):
        # npar = q + V = V(V+3)/2
        return int(round((-3 + math.sqrt(9 + 8 * npar)) / 2))

    def close(self):
        if self.group is not None:
            self.group.close()
        for ch in self.chains.values():
            ch.close()
        if self.comm is not None:
            self.comm.close()
            self.comm = None


# ------------------------------------------------------------------------------------------ chain driver
def initialize_and_run(X, y, c, total, V, R, eta, zeta, iota, aDelta, bDelta, nu, seed, prog_freq=0, purge_burn=None,
                       nsamp=None, callback=None, device=0):
    """initialize_and_run! (gibbs.jl:822-846) for ONE chain c; `seed` replaces the reference's rng argument
    (stream = seed + c).  Returns the Chain handle (the device-resident state table)."""
    tot_save = total if purge_burn is None else nsamp + purge_burn
    ch = Chain(X, y, R, tot_save, seed, c, device=device, eta=eta, zeta=zeta, iota=iota, aDelta=aDelta, bDelta=bDelta, nu=nu)
    ch.init_prior()
    nburn = total - nsamp
    run(ch, 2, nburn, total, purge_burn, prog_freq, callback if c == 1 else None)
    return ch


def run(chain, first_index, nburn, total, purge_burn=None, prog_freq=0, callback=None):
    """run! (gibbs.jl:849-864) on an existing chain."""
    return chain.run(first_index, nburn, total, purge_burn, prog_freq, callback)


def return_psrf_VOI(chainset, nburn, nsamp, fetch_state=True, summary_interval=None):
    """gibbs.jl:771-789: PSRF of gamma and xi over rows nburn+1..nburn+nsamp of every chain + chain 1's table.
    fetch_state=False leaves the table on the device (the top-up loops only look at the PSRF of the intermediate
    results); summary_interval=<credible level> adds the Summary statistics of chain 1 computed on the device."""
    rg, rx = chainset.rhat(nburn + 1, nsamp)
    state, dev = None, None
    if 1 in chainset.chains:
        ch = chainset.chains[1]
        if fetch_state:
            state = new_table(ch.tot, ch.V, ch.R, dead=True)
            ch.fetch(1, ch.tot, state)
        if summary_interval is not None:
            dev = device_summary(ch, nburn, nsamp, summary_interval)
    return Results(state, rx, rg, nburn, nsamp, dev)


def _finish(chainset, res, return_state, summary_interval, ess_max_lag=None):
    """The Results a fit returns: chain 1's table (states[1], gibbs.jl:788) and/or its Summary statistics from the device."""
    if ess_max_lag is not None:                       # collective over ranks, like the PSRF
        res.essgamma, res.essxi = chainset.ess(res.burn_in + 1, res.sampled, ess_max_lag if ess_max_lag > 0 else None)
    if 1 in chainset.chains:
        ch = chainset.chains[1]
        if return_state:
            res.state = new_table(ch.tot, ch.V, ch.R, dead=True)
            ch.fetch(1, ch.tot, res.state)
        if summary_interval is not None:
            res.summary_device = device_summary(ch, res.burn_in, res.sampled, summary_interval)
    return res


class _Progress:
    def __init__(self, total_ticks, enabled, start=0):
        self.n, self.total, self.enabled = start, max(total_ticks, 1), enabled

    def tick(self, _done=None):
        self.n += 1
        if self.enabled:
            sys.stderr.write("\rProgress: %3d%%" % min(100, int(100 * self.n / self.total)))
            sys.stderr.flush()

    def done(self):
        if self.enabled:
            sys.stderr.write("\n")


def _normalize_purge(purge_burn, nburn):
    """gibbs.jl:930-936."""
    if purge_burn is not None and purge_burn < nburn and purge_burn != 0:
        if nburn % purge_burn != 0:
            purge_burn = purge_burn - (nburn % purge_burn)
        return purge_burn
    return None


def generate_samples(X, y, R, eta=1.01, zeta=1.0, iota=1.0, aDelta=1.0, bDelta=1.0, nu=10, nburn=30000, nsamp=20000,
                     maxburn=50000, psrf_cutoff=1.2, x_transform=True, suppress_timer=False, num_chains=2, seed=None,
                     purge_burn=None, device=None, _keep=None, return_state=True, summary_interval=None, ess_max_lag=None):
    """generate_samples! (gibbs.jl:897-1020): "traditional" scheme with PSRF-driven top-up rounds."""
    if nu < R:
        pass                                       # the reference constructs an ArgumentError without throwing it (901-902)
    elif nu == R:
        print("Warning: ν==R may give poor accuracy. Consider increasing ν")
    X_new = XInput(X, x_transform)                 # X_new of gibbs.jl:907-918: element type kept, setup_X! runs on the device
    y = np.asarray(y, dtype=np.float64)
    total = nburn + nsamp
    prog_freq = 1000
    if prog_freq >= nburn:
        prog_freq = 10
    seed_eff = shared_seed(seed, lambda: random.SystemRandom().randrange(1, 2**31))     # Xoshiro() when seed===nothing; one draw per fit
    purge_burn = _normalize_purge(purge_burn, nburn)
    tot_save = total if purge_burn is None else nsamp + purge_burn
    hyper = dict(eta=eta, zeta=zeta, iota=iota, aDelta=aDelta, bDelta=bDelta, nu=nu)
    cs = ChainSet(X_new, y, R, num_chains, tot_save, seed_eff, hyper, device)
    if _keep is not None:
        _keep.append(cs)
    p = _Progress((total - 1) // prog_freq, not suppress_timer)
    cs.init_prior()
    cs.run(2, nburn, total, purge_burn, prog_freq, p.tick)
    p.done()
    tot_generated = nburn + nsamp
    stt = purge_burn if purge_burn is not None else nburn
    res = return_psrf_VOI(cs, stt, nsamp, fetch_state=False)
    print("%d samples generated. Max PSRF XI: %.2f. Max PSRF Gamma: %.2f" % (tot_generated, res.rhatxi.max(), res.rhatgamma.max()), file=sys.stderr)
    while (res.rhatxi.max() > psrf_cutoff or res.rhatgamma.max() > psrf_cutoff) and tot_generated < (maxburn + nsamp):
        # we want to generate nburn more samples (gibbs.jl:963-974)
        if purge_burn is not None:
            num2move = 1 if nsamp + purge_burn <= nburn else nsamp + purge_burn - nburn
        else:
            num2move = total - nburn
        tot_sze = tot_save
        print("num2move: %d nburn: %d nsamp: %d purge_burn: %s" % (num2move, nburn, nsamp, purge_burn), file=sys.stderr)
        p = _Progress((tot_generated + nburn - 1) // prog_freq, not suppress_timer, start=tot_generated // prog_freq)
        for ch in cs.chains.values():
            ch.move_rows(1, tot_sze - num2move + 1, num2move)                     # copy_table! loop :991-993
        cs.run(num2move + 1, (nburn - nsamp + num2move) if nburn > nsamp else 0,
               (num2move + nburn) if num2move > 1 else nburn, purge_burn, prog_freq, p.tick)   # run! :997-999
        p.done()
        A = (num2move + nburn) if num2move > 1 else nburn
        B = num2move
        tot_generated = tot_generated + A - B
        res = return_psrf_VOI(cs, stt, nsamp, fetch_state=False)
        print("%d samples generated. Max PSRF XI: %.3f. Max PSRF Gamma: %.3f" % (tot_generated, res.rhatxi.max(), res.rhatgamma.max()), file=sys.stderr)
    print("R = %s nu=%s nburn= %d nsamp = %d" % (R, nu, nburn, nsamp))
    print("%d samples generated. Max PSRF XI: %.3f. Max PSRF Gamma: %.3f\n" % (tot_generated, res.rhatxi.max(), res.rhatgamma.max()))
    res = _finish(cs, res, return_state, summary_interval, ess_max_lag)
    if _keep is None:
        cs.close()
    return res


def generate_samples_dbl(X, y, R, eta=1.01, zeta=1.0, iota=1.0, aDelta=1.0, bDelta=1.0, nu=10, mingen=10000,
                         maxgen=100000, psrf_cutoff=1.01, x_transform=True, suppress_timer=False, num_chains=2,
                         seed=None, purge_burn=None, device=None, return_state=True, summary_interval=None, ess_max_lag=None):
    """generate_samples_dbl! (gibbs.jl:1051-1198): "doubling generation" scheme."""
    if nu == R:
        print("Warning: ν==R may give poor accuracy. Consider increasing ν")
    nburn = _julia_round(mingen / 2)
    nsamp = mingen - nburn
    X_new = XInput(X, x_transform)
    y = np.asarray(y, dtype=np.float64)
    total = nburn + nsamp
    prog_freq = 1000
    if prog_freq >= nburn:
        prog_freq = 10
    seed_eff = shared_seed(seed, lambda: random.SystemRandom().randrange(1, 2**31))
    purge_burn = _normalize_purge(purge_burn, nburn)
    tot_save = total if purge_burn is None else nsamp + purge_burn
    hyper = dict(eta=eta, zeta=zeta, iota=iota, aDelta=aDelta, bDelta=bDelta, nu=nu)
    cs = ChainSet(X_new, y, R, num_chains, tot_save, seed_eff, hyper, device)
    p = _Progress((total - 1) // prog_freq, not suppress_timer)
    cs.init_prior()
    cs.run(2, nburn, total, purge_burn, prog_freq, p.tick)
    p.done()
    tot_generated = nburn + nsamp
    tot_samples = nsamp
    stt = purge_burn if purge_burn is not None else nburn
    res = return_psrf_VOI(cs, stt, nsamp, fetch_state=False)
    print("%d samples generated. Max PSRF XI: %.3f. Max PSRF Gamma: %.3f" % (tot_generated, res.rhatxi.max(), res.rhatgamma.max()), file=sys.stderr)

    def _bad(r):
        return (r.rhatxi.max() > psrf_cutoff or r.rhatgamma.max() > psrf_cutoff or np.isnan(r.rhatxi.max()) or np.isnan(r.rhatgamma.max()))

    while _bad(res) and tot_generated < maxgen:
        halfburn = _julia_round(mingen / 2)
        num2move = tot_samples                       # gibbs.jl:1143
        tot_samples = tot_samples + halfburn
        nsamp = tot_samples
        tot_sze = tot_save
        tot_save = tot_samples + halfburn
        print("num2move: %d nburn: %d nsamp: %d tot_save: %d first_index: %d" % (num2move, nburn, nsamp, tot_save, num2move + 1), file=sys.stderr)
        p = _Progress((tot_save - 1) // prog_freq, not suppress_timer, start=_julia_round((num2move + 1) / prog_freq))
        for ch in cs.chains.values():
            # new table of tot_save rows + copy_table!(state, states[c], 1:num2move, tail)  (gibbs.jl:1164-1172)
            ch.move_rows(1, tot_sze - num2move + 1, num2move)
            ch.resize(tot_save)
        cs.run(num2move + 1, 0, tot_save, purge_burn, prog_freq, p.tick)           # run! :1176-1178
        p.done()
        tot_generated = tot_generated + mingen
        res = return_psrf_VOI(cs, stt, nsamp, fetch_state=False)
        print("%d samples generated. Max PSRF XI: %.3f. Max PSRF Gamma: %.3f" % (tot_generated, res.rhatxi.max(), res.rhatgamma.max()), file=sys.stderr)
    print("\nR = %s nu=%s nburn= %d nsamp = %d\n" % (R, nu, nburn, nsamp))
    print("%d samples generated. Max PSRF XI: %.4f. Max PSRF Gamma: %.4f" % (tot_generated, res.rhatxi.max(), res.rhatgamma.max()))
    res = _finish(cs, res, return_state, summary_interval, ess_max_lag)
    cs.close()
    return res


def Fit(X, y, R, eta=1.01, V=30, zeta=1.0, iota=1.0, aDelta=1.0, bDelta=1.0, nu=10, nburn=30000, nsamples=20000,
        mingen=0, maxgen=0, psrf_cutoff=1.01, x_transform=True, suppress_timer=False, num_chains=2, seed=None,
        purge_burn=None, filename="parameters.log", device=None, return_state=True, summary_interval=None, ess_max_lag=None):
    """Fit! (gibbs.jl:725-751).  The `V` keyword is accepted and ignored, as in the reference.
    Extensions: summary_interval=95 computes Summary's statistics on the GPU (Results.summary_device);
    return_state=False then leaves the (large) state table on the device and frees it; ess_max_lag=0 (default lag
    window) or a lag count adds bulk effective sample sizes over all chains (Results.essgamma / essxi)."""
    seed = shared_seed(seed, lambda: random.randrange(1, 55556))          # sample(1:55555) :739; drawn on rank 0, the same on every rank
    if _rank_world()[0] == 0 and filename:
        with open(filename, "w") as f:
            f.write("BayesianNetworkRegression.jl Fit! function\n")
            f.write(datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S.%f")[:-3] + "\n")
            f.write(CITATION)
            f.write("\n\nParameters:\n")
            f.write("R=%s, η=%s, ζ=%s, ι=%s, aΔ=%s, bΔ=%s, ν=%s, nburn=%s, nsamples=%s, \n" % (R, eta, zeta, iota, aDelta, bDelta, nu, nburn, nsamples))
            f.write("mingen=%s, maxgen=%s, psrf_cutoff=%s, \n" % (mingen, maxgen, psrf_cutoff))
            f.write("x_transform=%s, suppress_timer=%s, num_chains=%s, purge_burn=%s \n" % (str(x_transform).lower(), str(suppress_timer).lower(), num_chains, "nothing" if purge_burn is None else purge_burn))
            f.write("seed=%s" % seed)
    if mingen > 0 and maxgen > 0:
        return generate_samples_dbl(X, y, R, eta=eta, zeta=zeta, iota=iota, aDelta=aDelta, bDelta=bDelta, nu=nu, mingen=mingen,
                                    maxgen=maxgen, psrf_cutoff=psrf_cutoff, x_transform=x_transform, suppress_timer=suppress_timer,
                                    num_chains=num_chains, seed=seed, purge_burn=purge_burn, device=device,
                                    return_state=return_state, summary_interval=summary_interval, ess_max_lag=ess_max_lag)
    return generate_samples(X, y, R, eta=eta, zeta=zeta, iota=iota, aDelta=aDelta, bDelta=bDelta, nu=nu, nburn=nburn, nsamp=nsamples,
                            maxburn=nburn + nsamples, psrf_cutoff=psrf_cutoff, x_transform=x_transform,
                            suppress_timer=suppress_timer, num_chains=num_chains, seed=seed, purge_burn=purge_burn, device=device,
                            return_state=return_state, summary_interval=summary_interval, ess_max_lag=ess_max_lag)
