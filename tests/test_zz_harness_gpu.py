"""Harness tests that start OTHER processes on the GPU box (bench.py under torch.distributed.run, rank workers).  They sort
after every parity test file on purpose: with `pytest -x` a harness failure can then never hide a parity test.  On failure
every rank's own stdout/stderr is part of the assertion message (torch.distributed.run's epilogue alone says nothing)."""
import glob
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import bnr_oracle as bo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_logs(log_dir, limit=6000):
    parts = []
    for f in sorted(glob.glob(os.path.join(log_dir, "**", "*.log"), recursive=True)):
        txt = open(f, errors="replace").read()
        parts.append("==== %s\n%s" % (os.path.relpath(f, log_dir), txt[-limit:]))
    return "\n".join(parts)


def _run_bench_ranks(tmp_path, nranks, extra, env_extra):
    log_dir = str(tmp_path / "ranks")
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "--redirects", "3", "--log-dir", log_dir, os.path.join(ROOT, "bench.py"), "--gpus", str(nranks)] + extra
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    logs = _rank_logs(log_dir)
    assert out.returncode == 0, "bench.py ranks failed (rc %d)\n%s\n==== launcher stderr (head)\n%s\n==== launcher stderr (tail)\n%s" % (
        out.returncode, logs, out.stderr[:1500], out.stderr[-1500:])
    # with --redirects 3 every rank's stdout lands in its log file; rank 0's holds the ONE JSON line
    lines = []
    for f in sorted(glob.glob(os.path.join(log_dir, "**", "stdout.log"), recursive=True)):
        lines += [ln for ln in open(f).read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, logs
    return json.loads(lines[0])


def test_bench_multi_rank_path_rehearsal(gpu, tmp_path):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank), rehearsed with two ranks on
    this one GPU (BNR_BENCH_ONE_DEVICE=1: both ranks use device 0, exchanges over gloo instead of RCCL): BASELINE configs[2]'s
    layout -- 8 chains IN ALL, chain c on rank (c-1) % 2, i.e. 4 per rank as one lockstep group -- rank 0 prints ONE JSON line whose
    value aggregates the chains of both ranks ("strong": the total work does not grow with N); the 8-chains-on-every-GPU figure is
    the weak_scaling sub-record."""
    d = _run_bench_ranks(tmp_path, 2, ["--steps", "40", "--warmup", "8", "--config", "cfg2", "--no-cpu-baseline"], dict(BNR_BENCH_ONE_DEVICE="1"))
    assert d["n_gpus"] == 2 and d["steps"] == 40 and d["warmup"] == 8 and d["scaling"] == "strong" and d["higher_is_better"] is True
    assert "8 chains total, 4 per GPU" in d["config"]["workload"] and d["config"]["chains_total"] == 8 and d["config"]["chains_per_gpu"] == 4
    assert d["value"] > 0 and d["roofline"]["achieved"] > 0
    assert abs(d["value"] - 8 * 40 / (d["ms_per_step"] * 40 / 1e3)) < 1e-6 * d["value"]
    assert d["counters"]["chol_fail"] == 0 and "single_chain" not in d
    w = d["weak_scaling"]
    assert w["scaling"] == "weak" and w["chains_per_gpu"] == 8 and w["chains_total"] == 16 and w["value"] > 0
    assert d["rhat_exchange"].startswith("bnr_rhat: host-callback"), d["rhat_exchange"]
    assert d["timed_region"]["sweeps_launched_eagerly"] == 0 and d["timed_region"]["sweeps_replayed_from_graphs"] == 40


def test_bench_plain_command_starts_two_ranks(gpu):
    """The driver's own spelling, `python bench.py --gpus 2 ...` with no launcher in front: bench.py starts the two ranks itself (fresh
    children, before the parent touched the GPU) and relays rank 0's line -- rehearsed on this one GPU (BNR_BENCH_ONE_DEVICE=1)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BNR_BENCH_ONE_DEVICE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "24", "--warmup", "4", "--config", "cfg2", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, "rc %d\n%s" % (out.returncode, out.stderr[-6000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 24 and d["scaling"] == "strong" and d["config"]["chains_per_gpu"] == 4 and d["value"] > 0
    assert d["timed_region"]["sweeps_replayed_from_graphs"] == 24


def test_bench_one_rank_rccl_path(gpu, tmp_path):
    """The N > 1 path of bench.py on hardware with ONE rank under torch.distributed.run: rendezvous over gloo, the library's own RCCL
    communicator (ncclCommInitRank of world size 1), the timing's max over ranks and bnr_rhat's exchange -- k_rhat_stats writes the
    per-chain messages into the communicator's device buffer, ncclAllGather (also with one rank the collective itself is called:
    the dlsym'd signature, the datatype code, the staging buffers and the stream are the ones eight ranks use), one copy to the
    host.  Everything of the N > 1 path except the peers."""
    d = _run_bench_ranks(tmp_path, 1, ["--steps", "24", "--warmup", "3", "--chains", "2", "--config", "cfg2", "--no-cpu-baseline"],
                         dict(BNR_BENCH_FORCE_DIST="1"))
    assert d["n_gpus"] == 1 and d["steps"] == 24 and d["value"] > 0 and d["max_rhat_gamma"] > 0 and d["config"]["chains_total"] == 2
    assert d["rhat_exchange"].startswith("bnr_rhat: ncclAllGather"), d["rhat_exchange"]
    assert d["timed_region"]["sweeps_launched_eagerly"] == 0 and d["timed_region"]["sweeps_replayed_from_graphs"] == 24
    assert 0 < d["roofline"]["sweep_frac"] < 1 and 0 < d["roofline"]["frac"] < 1


_RANK_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import bnr_amd
bnr_amd.lib(); bnr_amd.device_count()        # the library's HIP runtime comes up BEFORE torch maps its own copy (load-order guard of _capi.lib)
import torch.distributed as dist
rank = int(sys.argv[1]); os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[2]
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
dist.init_process_group("gloo", rank=rank, world_size=2)
d = np.load({data!r}); X, y = d["X"], d["y"]
keep = []
res = bnr_amd.generate_samples(X, y, 5, nburn=30, nsamp=20, maxburn=30, psrf_cutoff=1.2, x_transform=False, suppress_timer=True,
                               num_chains=3, seed=77, device=0, _keep=keep)
assert sorted(keep[0].chains) == ([1, 3] if rank == 0 else [2])
np.savez(sys.argv[3] + ".%d.npz" % rank, rg=res.rhatgamma, rx=res.rhatxi, has_state=np.array(res.state is not None),
         **{{"g%d" % c: ch.fetch(31, 50)["gamma"] for c, ch in keep[0].chains.items()}})
keep[0].close(); dist.barrier(); dist.destroy_process_group()
"""


def test_chains_sharded_over_two_ranks(gpu, tmp_path):
    """Two processes (one per rank, both on this box's GPU, gloo for the exchange): chains 1..3 are sharded round-robin,
    each rank samples its chains on the device, the per-chain Rhat messages are all-gathered and every rank finishes
    the same Rhat, equal to rhat() over all three chains."""
    port = _free_port()
    script = tmp_path / "rank.py"
    script.write_text(_RANK_WORKER.format(root=ROOT, data=os.path.join(G, "test1_xy.npz")))
    out = str(tmp_path / "o")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(port), out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed (rc %d)\n%s" % (r, p.returncode, "\n".join("==== rank %d\n%s" % (i, o[-6000:]) for i, o in enumerate(outs)))
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    assert np.array_equal(r0["rg"], r1["rg"]) and np.array_equal(r0["rx"], r1["rx"])
    assert bool(r0["has_state"]) and not bool(r1["has_state"])          # only chain 1's trace is returned (gibbs.jl:788)
    allg = np.stack([r0["g1"][:, :, 0], r1["g2"][:, :, 0], r0["g3"][:, :, 0]], axis=2)
    assert np.allclose(r0["rg"], bo.rhat(allg), rtol=1e-10)


def test_fit_in_a_fresh_process(gpu, tmp_path):
    """The README's first example in a process of its own: nothing loaded before `bnr_amd.Fit`, two chains as one lockstep group on the GPU,
    Summary on the device.  (Round 4: the chain-placement helpers imported torch, whose wheel maps its own HIP runtime -- the library's
    load-order guard then refused to create chains in any process that had not created one before.)"""
    code = ("import sys; sys.path.insert(0, %r); import numpy as np, bnr_amd\n"
            "X, y, _ = bnr_amd.make_synthetic(60, 8, 3, seed=5)\n"
            "res = bnr_amd.Fit(X, y, 3, V=8, nburn=40, nsamples=40, num_chains=2, seed=11, x_transform=False, suppress_timer=True, psrf_cutoff=50.0)\n"
            "g = [v for k, v in res.state.items() if np.asarray(v).shape[1:] == (36, 1)]\n"
            "print('OK', len(g) >= 2 and all(np.isfinite(np.asarray(v)).all() for v in g), 'torch' in sys.modules)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-4000:]
    assert out.stdout.strip().endswith("OK True False"), out.stdout[-2000:] + out.stderr[-2000:]

