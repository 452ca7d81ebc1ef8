#!/usr/bin/env python3
"""bench.py -- Gibbs iterations/s (all chains) of the MI355X hot path on BASELINE.json's headline workload.

  python bench.py --gpus N --steps K --warmup W            (N > 1 without a launcher: this script starts the N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one Gibbs sweep (gibbs_sample!, gibbs.jl:663-677) of every chain resident on the GPU.  Workload: synthetic
n=500, V=100 (q=5050), R=7 (SURVEY.md 8d generator, seed 20240501), the 8 chains of BASELINE.json configs[2] IN ALL:
chain c lives on rank (c-1) % N (the reference's pmap over chains, gibbs.jl:946-948), the 8/N chains of a GPU advance as one
lockstep group (one launch per kernel for all of them; N = 8: one chain per GPU, configs[2]'s literal layout).  Chains are
independent: no data-path collective; total work is fixed as N grows ("scaling": "strong").  After the timed region the
per-chain split-Rhat messages are all-gathered over RCCL (reported, not timed).  Sub-records of the same line: at N = 1
"single_chain" (one chain alone on the GPU, latency-bound), at N > 1 "weak_scaling" (8 chains on EVERY GPU).
Inputs are resident in HBM when the timed region starts.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    "cfg2": dict(n=200, V=50, R=5),
    "cfg3": dict(n=500, V=100, R=7),       # headline (BASELINE.json metric)
    "cfg4": dict(n=2000, V=200, R=7),
    "cfg5": dict(n=500, V=300, R=10),
}
FP64_MFMA_PEAK_TFLOPS = 78.6               # AMD MI355X FP64 matrix spec (not in the local guide)


_CPU_WORKER = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
import bnr_amd
from oracle import bnr_oracle as bo
n, V, R, seed, chain, budget, mode, threads = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), float(sys.argv[7]), int(sys.argv[8]), int(sys.argv[9])
blas = bo.use_numpy_openblas(threads) if mode == 2 else None
if mode == 2 and blas is None:
    mode = 1
X, y, _ = bnr_amd.make_synthetic(n, V, R, seed=seed)
tot = 4096                                # the time budget ends the run, not the table (<= ~70 iterations/s/chain)
o = bo.Oracle(X, y, R, tot, seed, chain=chain, pdf_mode=0, cost_mode=mode)
o.init_prior()
o.gibbs_sample(1, 2)                      # warm-up (page faults, thread pool)
t0 = time.time(); done = 1
while done < tot - 1 and (done < 3 or time.time() - t0 < budget):
    o.gibbs_sample(done + 1, done + 2); done += 1
print(done - 1, time.time() - t0, threads, mode, "|", blas or "")
"""


def _measured_mfma_peak():
    """TFLOP/s of a pure f64-MFMA loop on this chip, read from the committed microbenchmark output (never a literal)."""
    import re
    path = os.path.join(ROOT, "profiles", "round5_mfma_f64_peak.txt")
    try:
        vals = [float(m.group(1)) for line in open(path) if "(sustained)" in line for m in [re.search(r"([0-9.]+) TFLOP/s wall", line)] if m]
        return float(sorted(vals)[len(vals) // 2]) if vals else None
    except OSError:
        return None


def host_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (a GPU box shows all 256 hardware
    threads of its host, but its share per GPU is 16 cores -- oversubscribing spinning BLAS/OpenMP threads would understate the
    CPU baseline several-fold)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    cap = os.environ.get("BNR_BENCH_CPUS")
    return max(1, min(n, int(cap))) if cap else n


def cpu_baseline(n, V, R, seed, nchains, budget_s=15.0, mode=2):
    """The CPU oracle in reference-cost mode on the host cores: the same `nchains` chains as one OS process each (the reference's
    pmap workers, gibbs.jl:946), the host's hardware threads divided among them; iterations/s summed over the chains.
      mode 2: as the reference executes update_gamma! -- X/tau and Xt*tau2D copies, dense 2 n^2 q dgemm and LU solve by OpenBLAS
              (numpy's bundled libscipy_openblas: the reference's Julia links OpenBLAS too), dense (V-1)-dim pdfs;
      mode 1: the same operation counts with the oracle's own plain-C loops (OpenMP over the Gram columns), no BLAS."""
    import subprocess
    ncpu = host_cpus()
    per = max(1, min(64, ncpu // nchains))                  # numpy's OpenBLAS is built with MAX_THREADS = 64
    env = dict(os.environ, OMP_NUM_THREADS=str(per), OPENBLAS_NUM_THREADS=str(per))
    procs = [subprocess.Popen([sys.executable, "-c", _CPU_WORKER, ROOT, str(n), str(V), str(R), str(seed), str(c + 1), str(budget_s), str(mode), str(per)],
                              stdout=subprocess.PIPE, env=env, text=True) for c in range(nchains)]
    rate, its, threads, used, blas = 0.0, 0, 0, mode, ""
    for p in procs:
        head, _, tail = p.communicate(timeout=600)[0].partition("|")
        out = head.split()
        its += int(out[0]); rate += int(out[0]) / float(out[1]); threads += int(out[2]); used = int(out[3]); blas = tail.strip()
    how = ("Gram (dense 2n^2q dgemm) and LU solve by OpenBLAS: " + blas) if used == 2 else "plain-C loops of the oracle (OpenMP Gram, unblocked LU), no BLAS"
    return dict(value=rate, unit="Gibbs iterations/s (all chains)", cores=threads, kind="port",
                sample="%d chains x %.0f s wall (budget) of the same n/V/R workload (%d iterations in all), one process per chain with %d "
                       "threads each; reference-cost mode: %s" % (nchains, budget_s, its, per, how))


class _StdoutToStderr:
    """RCCL prints a banner (ROCm version, hostname, library path) on STDOUT when its first communicator is created; the
    contract of this script is ONE JSON line on stdout, so file descriptor 1 points at stderr while that happens."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def spawn_ranks(a):
    """`python bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): this process becomes the launcher -- it
    starts N fresh children (one rank per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, as torch.distributed.run would), BEFORE
    anything here has touched the GPU (nothing is imported but the standard library), relays rank 0's JSON line and returns the first
    non-zero exit code.  The reference's pmap spawns its workers itself too (gibbs.jl:946-948)."""
    import socket
    import subprocess
    with socket.socket() as so:                              # a free port on the loopback for the ranks' rendezvous
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL across processes)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    # every child is watched, not only rank 0: a rank that dies at start-up (bad device index, the library refusing to load) would leave the
    # others waiting in the rendezvous until torch's own timeout (~30 min) with the real error buried.  The first non-zero exit ends the run:
    # the remaining children (ours, addressed by pid) are terminated, that child's code is returned; the whole wait has an overall deadline.
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)     # rank 0's one line (drained so that its pipe never fills)
    reader.start()
    deadline = time.time() + float(os.environ.get("BNR_BENCH_DEADLINE_S", "3000"))
    rc, failed = 0, None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed, rc = bad[0]
            rc = rc if rc > 0 else 128 - rc             # a child killed by signal s reports -s
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            failed, rc = -1, 124
            break
        time.sleep(0.2)
    if failed is not None:
        sys.stderr.write("bench.py: %s; stopping the other ranks\n" % ("rank %d exited with code %d" % (failed, rc) if failed >= 0 else "the ranks did not finish before the deadline"))
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()                                      # (our own child, by its pid)
                p.wait()
    reader.join(timeout=10)
    out0 = "".join(c for c in chunks if c)
    sys.stdout.write(out0)
    sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--chains", type=int, default=8, help="chains of the fit in all (BASELINE configs[2]: 8), sharded round-robin over the ranks")
    ap.add_argument("--chains-per-gpu", type=int, default=0, help="diagnostics: this many chains on EVERY GPU instead (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=20240501)
    ap.add_argument("--overlap", type=int, default=1, help="0: single-stream schedule (diagnostics)")
    ap.add_argument("--graph-k", type=int, default=0, help="sweeps per captured graph (0: library default)")
    ap.add_argument("--graph", type=int, default=1, help="0: launch every kernel eagerly (diagnostics)")
    ap.add_argument("--binary-x", action="store_true", help="NOT the headline: the same shape with a 0/1 model matrix given as Bool (the reference's adjacency inputs), whose Gram "
                    "runs on the i8 matrix pipe (SURVEY 8f-2); the line says so in config.workload and roofline.kernel, dtype stays f64 (S, G and everything else are f64)")
    ap.add_argument("--dry-ranks", action="store_true", help="no GPU: every rank only reports which chains it would hold (checks the N-rank plumbing on a CPU)")
    a = ap.parse_args()

    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a))                           # the parent never touches the GPU: it only starts the ranks and relays rank 0's line
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks: the two must agree (one rank per GPU)" % (a.gpus, os.environ["WORLD_SIZE"]))
    if a.dry_ranks:
        world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
        total_chains = world * a.chains_per_gpu if a.chains_per_gpu > 0 else a.chains
        ids = [c for c in range(1, total_chains + 1) if (c - 1) % world == rank]
        if os.environ.get("BNR_BENCH_DRY_FAIL_RANK") == str(rank):      # test hook: this rank dies at start-up, before the rendezvous
            raise SystemExit(7)
        line = {"rank": rank, "world": world, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "chain_ids": ids, "chains_held": len(ids), "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"))}
        if world > 1:                                      # the ranks meet over gloo on the loopback, as the real run's rendezvous does
            import torch.distributed as dist
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            with _StdoutToStderr():                        # gloo reports its connections on stdout; the contract is ONE JSON line there
                dist.init_process_group("gloo", rank=rank, world_size=world)
                box = [None] * world
                dist.all_gather_object(box, line)
                dist.barrier()
                dist.destroy_process_group()
            line = {"dry_ranks": box, "n_gpus": world, "rccl_ranks": 0, "chains_held": [b["chains_held"] for b in sorted(box, key=lambda b: b["rank"])]}
        else:
            line = {"dry_ranks": [line], "n_gpus": 1, "rccl_ranks": 0, "chains_held": [line["chains_held"]]}
        if rank == 0:
            print(json.dumps(line))
        return

    import numpy as np
    import bnr_amd
    # The library binds the SYSTEM's HIP runtime and RCCL (/opt/rocm, ROCm 7.2): it is loaded before torch, whose wheel carries
    # its own libamdhip64/librccl (ROCm 7.0) -- whichever HIP runtime is loaded first serves the whole process.  torch is used
    # below ONLY as the rendezvous of the ranks (torch.distributed over gloo/TCP on the loopback: barriers, the broadcast of the
    # RCCL unique id); it never initialises the GPU.  Every device-side exchange runs on the library's own RCCL communicator.
    bnr_amd.lib()
    bnr_amd.device_count()                         # initialises that HIP runtime now (it must come up before torch's copies are loaded)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist, comm, exchange = None, None, "one process, no exchange"
    # rehearsal of the multi-rank path on a one-GPU box: every rank uses device 0 (RCCL refuses two ranks on one device: the
    # library communicator is then the host-callback kind, its all-gather carried by gloo)
    one_dev = os.environ.get("BNR_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    # BNR_BENCH_FORCE_DIST=1: take the multi-rank path also with ONE rank (rendezvous, RCCL communicator, barrier, all-gathers on
    # a one-GPU box: the hardware rehearsal of everything in the N > 1 path but the peers)
    if a.gpus > 1 or world > 1 or os.environ.get("BNR_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            # one node: neither the rendezvous (gloo) nor RCCL's bootstrap socket may depend on what the box's hostname or its
            # outward interface resolve to; the data path of RCCL stays xGMI / shared memory whatever carries the bootstrap
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        with _StdoutToStderr():                                 # RCCL prints a banner on stdout when a communicator comes up
            dist.init_process_group("gloo", rank=rank, world_size=world)
            if one_dev:
                comm = bnr_amd.make_comm(device=local_rank, force=True)
                exchange = "bnr_rhat: host-callback communicator over gloo (one-device rehearsal)"
            else:
                import torch
                why = ""
                try:
                    box = [bnr_amd.Comm.unique_id() if rank == 0 else None]
                except bnr_amd.BnrError as e:                   # librccl missing on rank 0: every rank must learn it
                    box, why = [None], str(e)
                dist.broadcast_object_list(box, src=0)
                if box[0] is not None:
                    try:
                        comm = bnr_amd.Comm.rccl(box[0], rank, world, local_rank)
                    except bnr_amd.BnrError as e:
                        why = str(e)
                # all ranks use the same transport: RCCL only if it came up everywhere (agreed over gloo)
                ok = torch.tensor([1 if comm is not None else 0])
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 1:
                    exchange = "bnr_rhat: ncclAllGather on the library's RCCL communicator (rendezvous: torch.distributed/gloo)"
                else:
                    if comm is not None:
                        comm.close()
                    comm = bnr_amd.make_comm(device=local_rank, force=True)
                    exchange = "bnr_rhat: host-callback communicator over gloo (the RCCL communicator did not come up on every rank%s)" % (": " + why if why else "")
            comm.allgather(np.zeros(8))                         # brings the channels up now, not in the timed region
            dist.barrier()

    def device_sync():
        bnr_amd.device_synchronize(local_rank)

    def max_over_ranks(v):
        return float(comm.allgather(np.array([v])).max()) if comm is not None else v

    cfg = CONFIGS[a.config]
    n, V, R = cfg["n"], cfg["V"], cfg["R"]
    q = V * (V + 1) // 2
    X, y, _truth = bnr_amd.make_synthetic(n, V, R, seed=a.seed)
    if a.binary_x:                                                # 0/1 adjacency-style data of the same shape, handed over as Bool (gibbs.jl:917: X_new keeps the element type)
        X = bnr_amd.XInput(np.asfortranarray(np.random.default_rng(a.seed).random((n, q)) < 0.5), False)
    K, W = a.steps, a.warmup
    total_chains = world * a.chains_per_gpu if a.chains_per_gpu > 0 else a.chains
    tot = W + K + max(100, min(K, 200)) + 1
    chains = []
    ids = [c for c in range(1, total_chains + 1) if (c - 1) % world == rank]  # round-robin over ranks, as api.local_chain_ids
    if not ids:
        raise SystemExit("bench.py: more ranks (%d) than chains (%d): rank %d would hold none" % (world, total_chains, rank))
    C = len(ids)                                                  # chains resident on THIS GPU
    for cid in ids:                                               # chain c uses stream seed + c (gibbs.jl:928)
        ch = bnr_amd.Chain(X, y, R, tot, a.seed, cid, device=local_rank) if not chains else bnr_amd.Chain.like(chains[0], a.seed, cid, tot)
        ch.init_prior()
        chains.append(ch)
    # several chains on one GPU advance in lockstep: one launch per kernel of a sweep covers all of them
    runner = bnr_amd.Group(chains) if C > 1 else chains[0]
    # what the N-rank run really was, as the transport reports it: ranks RCCL connected (ncclCommCount) and the chains every rank holds
    comm_info = comm.info() if comm is not None else {"kind": "none", "rank": 0, "world": 1, "rccl_ranks": 0, "rccl_rank": -1}
    held = comm.allgather(np.array([float(rank), float(local_rank), float(C), float(comm_info["rccl_ranks"])])) if comm is not None else np.array([[0.0, float(local_rank), float(C), 0.0]])
    if not a.overlap:
        runner.set_option("overlap", 0)
    if a.graph_k > 0:
        runner.set_option("graph_k", a.graph_k)
    if not a.graph:
        runner.set_option("graph", 0)

    def run_all(first, last, profile=False):
        if profile:
            runner.set_profiling(True)
        runner.run(first, last, last)
        if profile:
            runner.set_profiling(False)

    P = max(100, min(K, 200))                        # sweeps of the HIP-event pass around k_gram (after the timed region; at least 50 launches per schedule whatever --steps is:
                                                     # with the driver's 20 steps the pass used to average 10 eager launches, the first of them behind a schedule switch -- 207 us against
                                                     # the 190-191 us rocprofv3 reports over hundreds of launches of the same kernel)

    # capture + instantiate the hipGraphs here and replay them on scratch rows (discarded sweeps: the runtime's first-replay
    # setup and the clock ramp of an idle GPU stay out of the timed region whatever --warmup is); then the W warm-up steps
    for _ in range(int(os.environ.get("BNR_BENCH_PRIME", "2"))):
        runner.prepare()
    if W > 0:
        run_all(2, W + 1)
    device_sync()
    if dist:
        dist.barrier()
    device_sync()
    t0 = time.perf_counter()
    run_all(W + 2, W + K + 1)
    device_sync()
    if dist:
        dist.barrier()
    device_sync()
    dt = time.perf_counter() - t0
    eager_sweeps, replayed_sweeps = runner.last_timing(2)
    dt = max_over_ranks(dt)                                # MAX over ranks (an RCCL all-gather of one double per rank)

    # kernel-duration pass: the same sweeps continue, launched eagerly with HIP events recorded around every k_gram launch
    # on the stream it runs on (the timed region above replays captured graphs, where events cannot be read back)
    # first half: two-branch schedule as in the timed region (the events then also cover the time the launch waits for CUs
    # held by the concurrent scalar branch); second half: single-stream schedule, the events bracket the kernel alone --
    # that is the kernel duration the roofline uses (and what rocprofv3 --kernel-trace reports as its duration)
    P1 = P // 2
    run_all(W + K + 2, W + K + P1 + 1, profile=True)
    gram_us_pipe, _ = runner.last_timing(1)
    runner.set_option("overlap", 0)
    run_all(W + K + P1 + 2, W + K + P + 1, profile=True)
    runner.set_option("overlap", a.overlap)
    gram_us, gram_n = runner.last_timing(1)
    counters = chains[0].counters()

    # N = 1: one chain alone on the GPU (what every GPU does at N = 8; latency-bound).  N > 1: the weak-scaling figure, 8 chains on
    # EVERY GPU as one lockstep group.  Both continue from chain 1's data in the same process; reported as sub-records.
    single, weak, expected = None, None, None
    if C > 1 and world == 1:
        runner.close()
        runner = None
        Ks = min(max(K, 200), 1000)                   # (a sub-record with its own step count: at least 200 sweeps, so that the driver's short run reports the same one-chain rate as a long one)
        solo = bnr_amd.Chain.like(chains[0], a.seed, ids[0], Ks + 50)
        solo.init_prior()
        solo.run(2, 49, 49)
        device_sync()
        t1 = time.perf_counter()
        solo.run(50, Ks + 49, Ks + 49)
        device_sync()
        dts = max_over_ranks(time.perf_counter() - t1)
        single = {"value": world * Ks / dts, "unit": "iterations/s", "chains_per_gpu": 1, "steps": Ks, "ms_per_step": 1e3 * dts / Ks}
        solo.close()
        # What a SCALE record should be read against: the per-rank workload of the N-GPU layout -- the fit's chains sharded round-robin, total / N
        # chains per GPU as one lockstep group -- timed HERE on one GPU, times N.  Ranks never exchange anything inside the timed region (one all-gather per
        # convergence check, outside it), so N ranks of that workload are N copies of this measurement up to the slowest rank: a projection, not a measurement.
        expected = {"1": {"chains_per_gpu": C, "it_per_s_all_chains": total_chains * K / dt, "measured": "this line's value"}}
        for Nr in (2, 4, 8):
            if total_chains % Nr or total_chains // Nr < 1:
                continue
            Cr = total_chains // Nr
            if Cr == 1:
                per_rank = Ks / dts
            else:
                sub = [bnr_amd.Chain.like(chains[0], a.seed, 2000 + 10 * Nr + i, Ks + 50) for i in range(Cr)]
                for ch in sub:
                    ch.init_prior()
                sg = bnr_amd.Group(sub)
                sg.prepare()
                sg.run(2, 49, 49)
                device_sync()
                t1 = time.perf_counter()
                sg.run(50, Ks + 49, Ks + 49)
                device_sync()
                per_rank = Cr * Ks / (time.perf_counter() - t1)
                sg.close()
                for ch in sub:
                    ch.close()
            expected[str(Nr)] = {"chains_per_gpu": Cr, "it_per_s_per_gpu": per_rank, "it_per_s_all_chains": Nr * per_rank,
                                 "speedup_over_1_gpu": Nr * per_rank / (total_chains * K / dt), "measured": "per-rank workload on one GPU x %d" % Nr}
    elif world > 1 and a.chains_per_gpu == 0:
        if C > 1:
            runner.close()
        runner = None
        Kw, Cw = min(K, 1000), 8
        wch = [bnr_amd.Chain.like(chains[0], a.seed, 1000 + rank * Cw + i, Kw + 50) for i in range(Cw)]
        for ch in wch:
            ch.init_prior()
        wg = bnr_amd.Group(wch)
        wg.prepare()
        wg.run(2, 49, 49)
        device_sync()
        dist.barrier()
        t1 = time.perf_counter()
        wg.run(50, Kw + 49, Kw + 49)
        device_sync()
        dist.barrier()
        dtw = max_over_ranks(time.perf_counter() - t1)
        weak = {"value": world * Cw * Kw / dtw, "unit": "iterations/s", "scaling": "weak", "chains_per_gpu": Cw, "chains_total": world * Cw,
                "steps": Kw, "ms_per_step": 1e3 * dtw / Kw}
        wg.close()
        for ch in wch:
            ch.close()

    # convergence check over all chains of the job (return_psrf_VOI, gibbs.jl:771-789): bnr_rhat -- device reduction per chain, ONE
    # all-gather of the 4 (q + V)-double messages on the library's RCCL communicator (ncclAllGather over xGMI), Rhat finished on
    # every rank.  Not timed.
    from bnr_amd import _capi
    nsamp = K
    rh = None
    if nsamp >= 4:                                               # split-Rhat needs two samples per half
        rg, rx = _capi.rhat(chains, total_chains, comm, W + 1, nsamp)
        rh = np.concatenate([rg, rx])
    # effective sample size of the timed window over all chains (an addition to the reference's Rhat; same exchange pattern)
    ess = None
    if nsamp >= 64:
        Lag = min(250, nsamp // 4)
        local_e = {cid: ch.ess_stats(W + 2, nsamp, Lag) for cid, ch in zip(ids, chains)}
        stats_e = bnr_amd.allgather_stats(local_e, total_chains, comm) if comm is not None else np.stack([local_e[c] for c in sorted(local_e)])
        ess = bnr_amd.ess_from_stats(stats_e, nsamp, Lag)
    if comm is not None:
        comm.close()

    if rank == 0:
        value = total_chains * K / dt
        flops_gram = float(n) * n * q * C                         # algorithmic: symmetric X diag(S) X'  per chain of the launch
        traffic = None                                            # HBM bytes per k_gram launch from the PMC passes (tools/pmc_gram_round3.sh)
        pmc_file = os.path.join(ROOT, "profiles", "round6_gram_pmc.json")
        if not os.path.exists(pmc_file):
            pmc_file = os.path.join(ROOT, "profiles", "round5_gram_pmc.json")
        if a.config == "cfg3" and C in (1, 8) and not a.binary_x and os.path.exists(pmc_file):
            traffic = json.load(open(pmc_file))["k_gram<bnr_one,2>" if C == 1 else "k_gram8<bnr_many>"].get("traffic_bytes_per_launch")
        achieved = flops_gram / (gram_us * 1e-6) / 1e12 if gram_us > 0 else 0.0
        # whole-sweep roofline: F_iter = n^2 q + n^3/3 + 8 n q flops per chain-iteration, r = iterations/s per GPU
        f_iter = float(n) * n * q + float(n) ** 3 / 3.0 + 8.0 * n * q
        sweep_tflops = f_iter * (C * K / dt) / 1e12
        out = {
            "metric": "Gibbs iterations/sec (all chains)", "value": value, "unit": "iterations/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * dt / K, "higher_is_better": True,
            "scaling": "weak" if a.chains_per_gpu > 0 else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("synthetic n=%d V=%d (q=%d) R=%d, %d chains total, %s per GPU" % (n, V, q, R, total_chains, C if total_chains % world == 0 else "%d or %d" % (total_chains // world, total_chains // world + 1)))
                                   + (" -- BINARY 0/1 model matrix given as Bool (--binary-x; not the headline workload): Gram on the i8 matrix pipe" if a.binary_x else "")
                                   + ("" if world == 1 else " (BASELINE configs[2]: the %d chains of the fit sharded round-robin over %d GPUs, chain c on rank (c-1) %% %d; "
                                      "the chains of a GPU advance as one lockstep group)" % (total_chains, world, world)),
                       "chains_total": total_chains, "chains_per_gpu": C, "seed": a.seed},
            "timed_region": {"sweeps_replayed_from_graphs": int(replayed_sweeps), "sweeps_launched_eagerly": int(eager_sweeps)},
            "roofline": {"bound": "mfma", "kernel": ("k_sdigits + k_gram_i8 (X diag(S) X' of a 0/1 X: i8L exact v_mfma_i32_16x16x64_i8 Grams recombined in f64; achieved / frac are the f64-equivalent "
                                                      "algorithmic rate against the f64 peak, i.e. they may exceed 1)" if a.binary_x else "k_gram8 / k_gram (X diag(S) X', v_mfma_f64_16x16x4_f64)"), "achieved": achieved,
                         "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_source": "profiles/%s (FETCH_SIZE x2 + WRITE_SIZE of a launch of this shape, separate --pmc passes, tools/round6_profiles.sh / round5_profiles.sh)" % os.path.basename(pmc_file),
                         "sweep_frac": sweep_tflops / FP64_MFMA_PEAK_TFLOPS, "sweep_achieved": sweep_tflops, "sweep_flops_per_chain_iteration": f_iter,
                         "flops_per_launch": flops_gram, "avg_launch_us": gram_us, "launches_timed": gram_n,
                         "avg_launch_us_two_branch_schedule": gram_us_pipe,
                         "peak_measured_microbench": _measured_mfma_peak(),
                         "peak_measured_source": "profiles/round5_mfma_f64_peak.txt (tools/mfma_f64_peak.hip: pure v_mfma_f64_16x16x4_f64 loop, census-checked, median of the lines marked sustained); "
                                                 "the product kernel's K loop holds 2.39 GHz (median over its workgroups after 2.5 s of back-to-back launches, both clocks stamped in the kernel: profiles/round6_gram_clock.txt) -- the 2.10 GHz of "
                                                 "profiles/round5_gram_lab2_ablation.txt was a lab kernel's clock, not k_gram8's"},
            "max_rhat_gamma": None if rh is None else float(np.nanmax(rh[:q])), "max_rhat_xi": None if rh is None else float(np.nanmax(rh[q:])),
            "ess_gamma": None if ess is None else {"min": float(np.nanmin(ess[:q])), "median": float(np.nanmedian(ess[:q])),
                                                   "draws": int(nsamp * total_chains), "min_per_second": float(np.nanmin(ess[:q]) / dt)},
            "counters": counters, "rhat_exchange": exchange,
            "rccl_ranks": int(comm_info["rccl_ranks"]), "comm_kind": comm_info["kind"], "chains_held": [int(r[2]) for r in held],
            "ranks": [{"rank": int(r[0]), "local_rank": int(r[1]), "chains_held": int(r[2]), "rccl_ranks_seen": int(r[3])} for r in held],
        }
        if single is not None:
            out["single_chain"] = single
        if expected is not None:
            out["expected_scaling"] = expected
        if weak is not None:
            out["weak_scaling"] = weak
        if not a.no_cpu_baseline and world == 1:            # the host-core baseline is taken at N = 1 only
            out["cpu_baseline"] = cpu_baseline(n, V, R, a.seed, total_chains, 15.0, 2)
            out["cpu_baseline_no_blas"] = cpu_baseline(n, V, R, a.seed, total_chains, 6.0, 1)
        print(json.dumps(out))
    for ch in chains:
        ch.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
