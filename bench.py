#!/usr/bin/env python3
"""Headline benchmark: sliding-window solves/sec on the BASELINE.json workload
("10 kf, 200 pt, 40 line" + marginalisation prior = configs[2]) on N MI355X GPUs of one node.

One STEP = one pass of the hot path over one batch of B independent synthetic windows per GPU, already
resident in HBM: the fused solve kernel (8 fixed trust-region iterations, the per-frame budget regime of
estimator.cpp:1888-1897 made deterministic), the double2vector gauge fix (:1905) and the MARGIN_OLD marginalisation kernel
(estimator.cpp:1911-2046) that produces the next prior.  value = windows solved per second over all GPUs.

Multi-GPU: windows are independent (per-sequence replay shards one sequence per GPU), so every rank owns
B windows with rank-offset seeds and there is no data-path collective; RCCL is used only for the barrier
and the MAX-over-ranks time / SUM-over-ranks counts ("scaling": "weak").

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8 --steps 20 --warmup 3          # starts 8 fresh ranks itself (torch.distributed.run as a child, before any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        bench.py --gpus 8 --steps 20 --warmup 3             # what the driver runs: same ranks, launched from outside

Modes (`--mode`):
    solve   (default) the headline above; at N = 1 the line also carries the CPU baseline (all granted host cores + one core),
            the single-window latency and the PCIe-inclusive streaming rate as extra keys;
    replay  BASELINE configs[4]: `--streams` (default 8) EuRoC-trajectory streams, stream s on rank s mod N, every rank advances
            its streams in lock step through the native estimator (include/tcv_estimator.h): value = optimised windows per second;
            a step is one frame of every stream;
    stream  only the PCIe-inclusive figures: host-resident problems -> pack + H2D + solve + gauge fix + marginalisation + D2H,
            double-buffered over two HIP streams, and the end-to-end latency of a single window.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd"))
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default).  The lock-step replay uses two library streams per host thread (the
# marginalisation of a frame runs beside the next frame's association): with two host threads that is four streams next to the null stream,
# and two of them on one queue wait for each other's kernels.  Read by the runtime at its first call; an explicit setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0                 # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
SOLVER_ITERATIONS = 8
# ALGORITHMIC bytes per trust-region iteration per window (cfg 3: 200 pt, 40 line, prior n=75, L=50)
BYTES_PER_ITERATION_CFG3 = 89056
# ALGORITHMIC FP64 flops per trust-region iteration per window (cfg 3: assembly 1.41 M + landmark Schur 0.13 M + Cholesky and
# substitutions of the 171-dim system 1.73 M; the fused kernel does less -- chain elimination instead of the dense factorisation -- the figure
# prices the WORK of an iteration, like the bytes do)
FLOPS_PER_ITERATION_CFG3 = 3.3e6
DRY = os.environ.get("TCV_BENCH_DRY") == "1"      # tests only: ranks skip the device work (launch / rendezvous / reduction logic runs for real)


def algorithmic_bytes_per_iteration(n_imu, n_pt, n_ln, L, prior_n, prior_x0):
    """8*[S + N_imu*287 + N_pt*6 + N_ln*9 + 21 + (n^2 + n + X0)] + 4*[4N_imu + 4N_pt + N_ln] + 8*[(171+L) + 1]."""
    S = 77 + 99 + 7 + L
    prior = (prior_n * prior_n + prior_n + prior_x0) if prior_n else 0
    return 8 * (S + n_imu * 287 + n_pt * 6 + n_ln * 9 + 21 + prior) + 4 * (4 * n_imu + 4 * n_pt + n_ln) + 8 * ((171 + L) + 1)


def csrc_sha16() -> str:
    """hash of the kernel sources (tools/profile_round.sh stores the same in profiles/counters.json): static counter figures are printed only
    next to the sources they were measured on (the GPU box has no .git to ask)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "tc-viml_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]



def fuse(b):
    """the gauge fix in the solve kernel's epilogue for every batch this file times (tcv_batch_set_fused_gauge_fix: gauge_fix() behind the solve is then a
    no-op; TCV_BENCH_SEPARATE_GAUGE=1: the stand-alone kernel of rounds 1-5, same bits)"""
    if not os.environ.get("TCV_BENCH_SEPARATE_GAUGE"):
        b.fuse_gauge_fix()
    return b

def shard_ids(rank: int, per_gpu: int) -> int:
    """first synthetic window id of a rank: disjoint seeds per rank, fixed work per GPU (weak scaling)."""
    return 100000 + rank * per_gpu


def shard_streams(n_streams: int, rank: int, world: int):
    """sequence s -> GPU s mod G."""
    return [s for s in range(n_streams) if s % world == rank]


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n_gpus: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: the parent -- which has not touched the GPU -- starts N fresh ranks as a CHILD
    process (torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1) and returns its exit code.  Never an exec of a
    process that initialised the GPU."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def dist_setup(n_gpus: int, backend: str | None = None):
    """(rank, world, local_rank, dist-or-None).  For N > 1 every rank is one process launched by torch.distributed.run."""
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != n_gpus:
        raise SystemExit(f"bench.py: --gpus {n_gpus} but WORLD_SIZE={world}: launch exactly one rank per GPU "
                         f"(python bench.py --gpus {n_gpus} starts them itself)")
    if world == 1:
        return 0, 1, 0, None
    if not os.environ.get("TCV_BENCH_NO_PIN"):
        cpu_share(local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))      # (before torch / the library start their threads)
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or ("gloo" if DRY or not torch.cuda.is_available() else "nccl")
    if backend == "nccl":
        torch.cuda.set_device(local)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local, dist


def reduce_stats(dist, elapsed_s: float, windows: int, device=None, extra=()):
    """MAX of the elapsed time and SUM of the counts over ranks (the only collectives of the whole job; SURVEY.md 8(e):
    one all-reduce of {solves, iterations, elapsed})."""
    if dist is None:
        return (elapsed_s, windows) + tuple(extra) if extra else (elapsed_s, windows)
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    w = torch.tensor([float(windows)] + [float(v) for v in extra], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(w, op=dist.ReduceOp.SUM)
    out = (float(t.item()), int(w[0].item()))
    return out + tuple(int(v) for v in w[1:].tolist()) if extra else out


def gather_rank_rates(dist, elapsed_s: float, windows: int, device=None):
    """per-rank windows/s as every rank measured it (rank order): a straggler is visible on the line"""
    if dist is None:
        return [windows / elapsed_s if elapsed_s > 0 else 0.0]
    import torch
    mine = torch.tensor([float(windows), float(elapsed_s)], dtype=torch.float64, device=device)
    got = [torch.zeros(2, dtype=torch.float64, device=device) for _ in range(dist.get_world_size())]
    dist.all_gather(got, mine)
    return [float(g[0].item()) / float(g[1].item()) if float(g[1].item()) > 0 else 0.0 for g in got]


def cpu_share(local: int, n_local: int):
    """the host cores of one rank when several ranks share a node: the process's affinity mask cut into n_local contiguous pieces.  Set
    before anything starts a thread (the library's worker threads inherit it, and size their pool by it), so the ranks' host sides do not
    migrate over each other's cores.  Returns the share, or None when there is nothing to cut."""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    if n_local <= 1 or len(cores) < n_local:
        return None
    per = len(cores) // n_local
    share = cores[local * per:(local + 1) * per]
    try:
        os.sched_setaffinity(0, share)
    except OSError:
        return None
    return share


def ranks_seen(dist, device=None) -> int:
    """number of ranks the collective backend (RCCL on the GPU box) actually connects: all-reduce of 1."""
    if dist is None:
        return 1
    import torch
    one = torch.ones(1, dtype=torch.float64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(one.item())


def build_batches(tcv, synth, first_id: int, B: int):
    """pre-window batch (no prior) -> GPU solve + marginalise -> priors -> main-window batch with priors and its own
    marginalisation problems.  Everything is produced by the HIP path; nothing here touches oracle/."""
    opts = tcv.default_options(SOLVER_ITERATIONS, True)
    pre = synth.make_windows(first_id, B, frame_shift=-1)
    pw = [synth.window_at(pre, k) for k in range(B)]
    W = [tcv.Window(w) for w in pw]
    MW = [tcv.margin_old_window(w) for w in pw]
    M = [tcv.Window(mw, share=W[k]) for k, mw in enumerate(MW)]
    drops = [tcv.margin_old_drops(W[k], MW[k]) for k in range(B)]
    b0 = tcv.Batch(W, M, drops)
    b0.solve(opts); b0.marginalize(); b0.synchronize(); b0.download_priors()
    pdev = b0.priors_device()          # the same priors as device-resident handles (their numbers stay in b0's result buffer): --mode stream
    main = synth.make_windows(first_id, B)
    wins, keep = [], []
    for k in range(B):
        P = b0.prior(k)
        d = P.export(); d["blocks"] = tcv.shifted_prior_blocks(P, W[k])
        w = dict(synth.window_at(main, k)); w["prior"] = d
        wins.append(w)
    Wm = [tcv.Window(w) for w in wins]
    MWm = [tcv.margin_old_window(w) for w in wins]
    Mm = [tcv.Window(mw, share=Wm[k], prior=Wm[k].prior) for k, mw in enumerate(MWm)]
    dropsm = [tcv.margin_old_drops(Wm[k], MWm[k]) for k in range(B)]
    batch = tcv.Batch(Wm, Mm, dropsm)
    return batch, wins, (Wm, Mm, dropsm, pdev)


# ---- CPU baseline (the ONLY part of this file that touches oracle/) -------------------------------------------------------
def _cpu_worker(args):
    """one host process = one window stream, one thread like Ceres num_threads = 1: solve (8 fixed iterations) + MARGIN_OLD
    marginalisation per window with the C restatement (oracle/tcv_oracle.c)."""
    wins, budget_s = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    orc.lib()
    n, t_solve = 0, 0.0
    t_end = time.perf_counter() + budget_s
    while time.perf_counter() < t_end:
        ws = [orc.Window(w) for w in wins]          # fresh copies (the solve updates the states in place); not timed
        t0 = time.perf_counter()
        for O in ws:
            O.solve(SOLVER_ITERATIONS, True)
            O.marginalize_old()
        t_solve += time.perf_counter() - t0
        n += len(ws)
    return n, t_solve


def _cpu_quota():
    """cores' worth of CPU time the cgroup grants (cpu.max), or None."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        return None


def cpu_pool():
    """worker processes of the all-core CPU baseline, forked BEFORE this process touches the GPU (they idle until the windows
    exist).  One process per granted core (affinity mask, capped by the cgroup quota rounded up and by 64)."""
    import multiprocessing as mp
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    q = _cpu_quota()
    if q:
        n = min(n, max(1, int(math.ceil(q))))
    n = max(1, min(n, 64))
    return mp.get_context("fork").Pool(n), n


def cpu_baseline(wins, pool, nproc, budget_s: float = 12.0):
    """the CPU restatement on a bounded sample of the SAME windows -- on every granted host core (one process
    per window stream) and on one core."""
    per = 4
    t0 = time.perf_counter()
    res = pool.map(_cpu_worker, [([wins[(i * per + j) % len(wins)] for j in range(per)], budget_s) for i in range(nproc)])
    wall = time.perf_counter() - t0
    pool.close(); pool.join()
    tot = sum(r[0] for r in res); tmax = max(r[1] for r in res)
    rates = sorted(r[0] / r[1] for r in res)          # per process: its solves over ITS time inside the solver
    one_n, one_t = _cpu_worker(([wins[j % len(wins)] for j in range(per)], min(4.0, budget_s)))
    flags = "unknown flags"
    try:
        for line in open(os.path.join(ROOT, "oracle", "Makefile")):
            if line.startswith("CFLAGS"):
                flags = "gcc " + line.split("=", 1)[1].strip()
    except OSError:
        pass
    # value = the sum of the per-process rates (what the box delivers with every granted core busy); single_core_value = the median
    # per-process rate of THAT run, so that value / cores and single_core_value are the same measurement seen two ways; the rate of one
    # process alone on the otherwise idle box (higher or lower: boost clocks, the parent's GPU threads) is reported next to it
    return {"value": sum(rates), "unit": "solves/s", "cores": nproc, "kind": "port",
            "sample": f"{tot} solves of the benchmark's windows ({SOLVER_ITERATIONS} fixed iterations + 1 MARGIN_OLD marginalisation each) on {nproc} host "
                      f"processes for {tmax:.1f} s ({wall:.1f} s wall), one thread per window stream like Ceres num_threads = 1 "
                      f"(oracle/tcv_oracle.c, {flags}; dense Schur, not Ceres)",
            "single_core_value": rates[len(rates) // 2], "single_core_sample": f"median per-process rate of the {nproc}-process run (min {rates[0]:.1f}, max {rates[-1]:.1f})",
            "one_process_alone_value": one_n / one_t, "one_process_alone_sample": f"{one_n} solves in {one_t:.1f} s, one process after the others have finished",
            "host_cpu": _cpu_model(), "host_cores_available": os.cpu_count(), "cgroup_cpu_quota_cores": _cpu_quota()}


def _cpu_replay_worker(args):
    """one host process = one EuRoC-trajectory stream replayed through the Python window management with the CPU ORACLE back end
    (tests/replay_oracle.py: the C restatement solves and marginalises every window, one thread like Ceres num_threads = 1).  Timed: the
    back end's optimize() calls only (solve + gauge fix + marginalisation of a window) -- NOT the Python window management, the NumPy
    pre-integration or the NumPy 2D-3D association around them, which the reference does in C++."""
    stream_id, n_frames, features, lines, budget_s = args
    for d in ("oracle", "tests", "tc-viml_amd"):
        sys.path.insert(0, os.path.join(ROOT, d))
    import replay
    from replay_oracle import OracleBackend

    class Timed(OracleBackend):
        t = 0.0
        n = 0
        t_end = None

        def optimize(self, win, marg_flag, num_iterations, fixed_iterations):
            if self.t_end is not None and time.perf_counter() > self.t_end:
                raise TimeoutError
            t0 = time.perf_counter()
            r = OracleBackend.optimize(self, win, marg_flag, num_iterations, fixed_iterations)
            self.t += time.perf_counter() - t0
            self.n += 1
            return r

    seqs = list(replay.EUROC_SEQUENCES)
    st = replay.simulate_stream_euroc(seqs[stream_id % len(seqs)], n_frames, start_s=(0.5 + 3.0 * (stream_id // len(seqs))) % max(3.0, 33.0 - 0.1 * n_frames),
                                      max_features=features, max_lines=lines, associate=True)
    be = Timed()
    be.t_end = time.perf_counter() + budget_s
    try:
        replay.run(st, be, num_iterations=SOLVER_ITERATIONS)
    except TimeoutError:
        pass
    return be.n, be.t


def cpu_replay_baseline(pool, nproc, n_frames, features, lines, budget_s, close=True):
    """the replay workload's CPU figure: every granted core replays one of the SAME streams (stream i on process i) for a bounded time"""
    t0 = time.perf_counter()
    res = pool.map(_cpu_replay_worker, [(i, n_frames, features, lines, budget_s) for i in range(nproc)], chunksize=1)
    wall = time.perf_counter() - t0
    if close:
        pool.close(); pool.join()
    res = [r for r in res if r[0] > 0 and r[1] > 0]
    if not res:
        return None
    rates = sorted(n / t for n, t in res)
    return {"value": sum(rates), "unit": "solves/s", "cores": len(rates), "kind": "port",
            "sample": f"{sum(n for n, _ in res)} windows of the replay's own streams (stream i on process i, {features} features + {lines} line tracks per frame, association "
                      f"in the loop) solved and marginalised by the C restatement (oracle/tcv_oracle.c, one thread per stream like Ceres num_threads = 1; dense Schur, "
                      f"not Ceres) in {max(t for _, t in res):.1f} s of back-end time per process ({wall:.1f} s wall: the Python window management, NumPy pre-integration "
                      f"and NumPy association around the back end are NOT in the rate)",
            "single_core_value": rates[len(rates) // 2], "single_core_sample": f"median per-process rate (min {rates[0]:.1f}, max {rates[-1]:.1f})",
            "host_cpu": _cpu_model(), "host_cores_available": os.cpu_count(), "cgroup_cpu_quota_cores": _cpu_quota()}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ---- PCIe-inclusive figures ---------------------------------------------------------------------------------------------
def stream_figures(tcv, torch, keep, B_stream: int = 512, rounds: int = 4, wins=None, local: int = 0):
    """host-resident problems -> tcv_batch_create (pack + H2D) -> solve -> gauge fix -> marginalisation -> D2H of states and
    priors -> destroy, overlapped: two to four host threads, each with its own windows and HIP stream, so that one batch packs / copies
    while another computes.  Also the end-to-end and kernel-only latency of ONE window (the real-time single-estimator case)."""
    import threading
    Wm, Mm, dropsm, pdev = keep
    # host threads, each with its own windows and HIP stream: as many 512-window sets as the batch holds, between 2 and 4
    # (`--mode stream --windows 2048`: four)
    NTH = int(os.environ.get("TCV_STREAM_THREADS", str(max(2, min(4, len(Wm) // B_stream)))))
    B = min(B_stream, len(Wm) // NTH)
    opts = tcv.default_options(SOLVER_ITERATIONS, True)
    # (the argument arrays of tcv_batch_create -- handles and drop lists -- are built once per window set: a C++ caller has them at hand)
    halves = [tcv.BatchSpec(Wm[i * B:(i + 1) * B], Mm[i * B:(i + 1) * B], dropsm[i * B:(i + 1) * B]) for i in range(NTH)]
    stage = {"pack_h2d": 0.0, "compute": 0.0, "d2h": 0.0}

    resident = [True]      # priors device-resident (the default path since round 4) or through the host (round 3's path, for comparison)

    def one_pass(h, stream_ptr, acc=None):
        t0 = time.perf_counter()
        b = fuse(tcv.Batch(None, spec=h) if isinstance(h, tcv.BatchSpec) else tcv.Batch(*h))
        t1 = time.perf_counter()
        b.solve(opts, stream_ptr); b.gauge_fix(stream_ptr); b.marginalize(stream_ptr); b.synchronize()
        t2 = time.perf_counter()
        b.download_states()
        if resident[0]:
            # last_marginalization_info stays in HBM: per window a handle (layout on the host) and two ints come down; the next frame's
            # tcv_batch_create would splice it device-to-device exactly like the priors this pass was created from
            pri = b.priors_device_raw()
            t3 = time.perf_counter()
            tcv.lib().tcv_priors_destroy(pri, len(pri))
        else:
            b.download_priors(compact=True)
            pri = b.priors()      # host-resident tcv_prior of every window (J0, r0, linearisation point): what the next frame's problem takes
            t3 = time.perf_counter()
            del pri
        del b
        if acc is not None:
            acc["pack_h2d"] += t1 - t0; acc["compute"] += t2 - t1; acc["d2h"] += t3 - t2

    def worker(h, st):
        tcv.check(tcv.lib().tcv_set_device(local))      # the current HIP device is per host thread
        for _ in range(rounds):
            one_pass(h, st)

    # (TCV_STREAM_THREAD: every host thread launches on its own library stream, the one its uploads and downloads use anyway -- one stream
    # per thread instead of two on the runtime's four hardware queues)
    streams = [tcv.STREAM_THREAD for _ in range(NTH)]

    def timed():
        for k in stage:
            stage[k] = 0.0
        one_pass(halves[0], streams[0])          # warm-up
        for _ in range(2):
            one_pass(halves[0], streams[0], stage)
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(halves[i], streams[i])) for i in range(NTH)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return time.perf_counter() - t0, dict(stage)

    # round 3's path first (host priors: pack + upload 46 KB, download 62 KB per window), then the same windows with their priors as
    # device-resident handles (same bits in the pool: tests/test_gpu_resident.py)
    resident[0] = False
    dt_host, stage_host = timed()
    nbind = NTH * B
    tcv.check(tcv.lib().tcv_problems_set_marginalization_prior((C.c_void_p * nbind)(*[w.h for w in Wm[:nbind]]), (C.c_void_p * nbind)(*[p.h for p in pdev[:nbind]]), nbind))
    tcv.check(tcv.lib().tcv_problems_set_marginalization_prior((C.c_void_p * nbind)(*[w.h for w in Mm[:nbind]]), (C.c_void_p * nbind)(*[p.h for p in pdev[:nbind]]), nbind))
    resident[0] = True
    dt, stage = timed()
    # one window: kernel-only (resident) and end-to-end (host blocks -> batch -> solve -> gauge fix -> marginalisation -> states and prior back).
    # The window is built afresh from its INITIAL states: the passes above left the batch's windows solved (download_states writes into the
    # caller's blocks, like ceres::Solve), and a converged window rejects its steps, i.e. skips most linearisations (round 2 timed that: 1.18 ms).
    def fresh_one():
        if wins is None:
            return ([Wm[0]], [Mm[0]], [dropsm[0]])
        # (steady state of an estimator: the incoming prior is the previous frame's device-resident result)
        W1 = tcv.Window(wins[0], prior=pdev[0]); mw = tcv.margin_old_window(wins[0]); M1 = tcv.Window(mw, share=W1, prior=W1.prior)
        return ([W1], [M1], [tcv.margin_old_drops(W1, mw)])
    one = fresh_one()
    b1 = fuse(tcv.Batch(*one))
    co1 = b1.cooperative()
    lat = []
    for _ in range(12):
        t1 = time.perf_counter()
        b1.solve(opts); b1.gauge_fix(); b1.marginalize(); b1.synchronize()      # (resident: every repetition starts from the batch's initial states)
        lat.append(time.perf_counter() - t1)
    st1 = b1.stats()
    e2e = []
    for _ in range(6):
        one = fresh_one()
        t1 = time.perf_counter()
        one_pass(one, None)
        e2e.append(time.perf_counter() - t1)
    # what the native estimator's caller waits for since round 4: host blocks -> batch -> solve -> gauge fix -> states on the host; the
    # marginalisation is launched behind them and its prior handed on without a wait (tcv_batch_get_priors_device_async)
    e2s = []
    for _ in range(6):
        one = fresh_one()
        t1 = time.perf_counter()
        bq = fuse(tcv.Batch(*one))
        bq.solve(opts); bq.gauge_fix(); bq.download_states()
        e2s.append(time.perf_counter() - t1)
        bq.marginalize(); keepq = bq.priors_device(nowait=True)
        bq.synchronize()
        del keepq, bq
    return {"stream_solves_per_s": NTH * rounds * B / dt,
            "stream_host_priors_solves_per_s": NTH * rounds * B / dt_host,
            "stream_note": f"PCIe-inclusive: {NTH} host threads x {rounds} passes x {B} windows, per pass pack + H2D (tcv_batch_create), solve + gauge fix + "
                           f"marginalisation on the thread's own HIP stream, D2H of the states; the marginalisation priors are DEVICE-RESIDENT "
                           f"(tcv_batch_get_priors_device: made on the device, spliced device-to-device into every pass's pool, the pass's own new "
                           f"priors taken as handles: two ints per window down); serial stage times per {B}-window pass [ms]: "
                           + ", ".join(f"{k} {1e3 * v / 2:.1f}" for k, v in stage.items())
                           + "; stream_host_priors_solves_per_s = round 3's path (priors packed + uploaded, downloaded as 62 KB per window and "
                           "turned into host objects): " + ", ".join(f"{k} {1e3 * v / 2:.1f}" for k, v in stage_host.items()),
            "single_window_ms": {"resident_launch_to_sync": 1e3 * float(np.median(lat)), "kernels": st1["solve_ms"] + st1["marg_ms"],
                                 "solve_kernel": st1["solve_ms"], "marg_kernel": st1["marg_ms"], "host_blocks_end_to_end": 1e3 * float(np.median(e2e)),
                                 "host_blocks_to_states": 1e3 * float(np.median(e2s)),
                                 "workgroups_per_window": 1 + co1["helpers"],
                                 "note": "one cfg-3 window from its initial states, 8 full iterations (cooperative small-batch kernels); host_blocks_to_states: "
                                         "until the solved, gauge-fixed states are on the host -- the marginalisation runs behind them, off the caller's path "
                                         "(tcv_batch_get_priors_device_async), as in the native estimator"}}


def sweep_figures(tcv, keep, opts):
    """B in {256, 1024, 4096} windows per launch and the run-to-convergence rate -- kernel-side (solve + gauge fix +
    marginalisation launches, launch to sync), three launches each.  The 4096-window batch holds the benchmark's 1024 windows four times
    (every entry is packed into its own slice of the pool: the same traffic as 4096 different windows of this shape)."""
    Wm, Mm, dropsm, _pdev = keep
    n0 = len(Wm)
    res = {}
    for B in (256, 1024, 4096):
        if n0 < min(B, 1024):
            continue
        idx = [k % n0 for k in range(B)]
        b = fuse(tcv.Batch([Wm[k] for k in idx], [Mm[k] for k in idx], [dropsm[k] for k in idx]))
        b.solve(opts); b.gauge_fix(); b.marginalize(); b.synchronize()          # warm-up
        t0 = time.perf_counter()
        for _ in range(3):
            b.solve(opts); b.gauge_fix(); b.marginalize()
        b.synchronize()
        dt = (time.perf_counter() - t0) / 3
        st = b.stats()
        res[str(B)] = {"solves_per_s": B / dt, "ms_per_launch_set": 1e3 * dt, "solve_kernel_ms": st["solve_ms"], "marg_kernel_ms": st["marg_ms"]}
        if B == min(1024, n0):
            # run to convergence: Ceres' tolerances decide (function 1e-6, gradient 1e-10, parameter 1e-8), at most 50 iterations (Ceres' default cap)
            oc = tcv.default_options(50, False)
            b.solve(oc); b.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                b.solve(oc); b.gauge_fix(); b.marginalize()
            b.synchronize()
            dtc = (time.perf_counter() - t0) / 3
            sm = b.summaries()
            its = np.array([sm[k].num_iterations for k in range(B)])
            res_conv = {"solves_per_s": B / dtc, "ms_per_launch_set": 1e3 * dtc, "windows": B, "max_num_iterations": 50,
                        "iterations_mean": float(its.mean()), "iterations_median": float(np.median(its)), "iterations_max": int(its.max()),
                        "note": "the same windows solved until Ceres' convergence tests fire (estimator.cpp:1888-1897 sets only the iteration cap and the time budget; "
                                "tolerances are Ceres' defaults) + gauge fix + marginalisation; a launch lasts as long as its slowest window"}
        del b
    return {"batch_sweep": res, "run_to_convergence": res_conv if res else None}


# ---- modes ----------------------------------------------------------------------------------------------------------------
def run_solve(args, rank, world, local, dist):
    pool = nproc = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not DRY:
        pool, nproc = cpu_pool()                     # forked before the first GPU call
    B = args.windows
    if DRY:
        torch = tcv = None
        dev = None
        step = lambda: time.sleep(0.001)
        sync = (lambda: dist.barrier()) if dist is not None else (lambda: None)
        wins = None
    else:
        import torch
        import synth
        import tcv
        if tcv.lib().tcv_device_count() < 1:
            raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
        tcv.check(tcv.lib().tcv_set_device(local))
        tcv.check(tcv.lib().tcv_set_solver_variant(args.variant))
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        batch, wins, keep = build_batches(tcv, synth, shard_ids(rank, B), B)
        opts = tcv.default_options(SOLVER_ITERATIONS, True, True, args.threads)

        fuse(batch)                    # double2vector() in the solve kernel's epilogue; gauge_fix() below is then a no-op

        def step():
            batch.solve(opts)
            batch.gauge_fix()          # double2vector(), estimator.cpp:1905: the marginalisation linearises at the gauge-fixed states
            batch.marginalize()

        def sync():
            batch.synchronize()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()

    red_dev = dev if (dist is not None and dist.get_backend() == "nccl") else None
    n_ranks = ranks_seen(dist, red_dev)
    for _ in range(args.warmup):
        step()
    sync()
    solve_ms, marg_ms = [], []
    t0 = time.perf_counter()
    sample_every = max(1, (args.steps + 63) // 64)      # HIP-event duration of the launches (events recorded on the launch stream): at most 64 samples
    for k in range(args.steps):
        step()
        if not DRY and k % sample_every == 0:
            batch.synchronize()
            st = batch.stats(); solve_ms.append(st["solve_ms"]); marg_ms.append(st["marg_ms"])
    sync()
    elapsed = time.perf_counter() - t0
    elapsed_max, windows_total, iters_total = reduce_stats(dist, elapsed, B * args.steps, red_dev, extra=(B * args.steps * SOLVER_ITERATIONS,))
    rank_rates = gather_rank_rates(dist, elapsed, B * args.steps, red_dev)

    if not DRY:
        # parity of what was just timed is the job of tests/ and smoke(); here only sanity of the results
        s = batch.summaries(min(4, B))
        assert all(np.isfinite(s[k].final_cost) and s[k].final_cost < s[k].initial_cost for k in range(min(4, B)))

    if rank != 0:
        return
    out = {
        "metric": "sliding-window solves/sec (10 kf, 200 pt, 40 line)",
        "value": windows_total / elapsed_max,
        "unit": "solves/s",
        "n_gpus": n_ranks,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed_max / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "configs[2]: synthetic 10-kf window, 200 point + 40 2D-3D line residual blocks + marginalisation prior "
                               "(n=75), 50 landmarks; 8 fixed dogleg iterations + 1 MARGIN_OLD marginalisation per solve",
                   "windows_per_gpu": B, "solver_iterations": SOLVER_ITERATIONS, "parallelism": f"independent windows x{world}",
                   "threads_per_window": args.threads, "layout": "chain" if args.variant == 0 else "dense"},
        "iterations_per_s": iters_total / elapsed_max,
        "ranks_seen": n_ranks,
        "per_rank_solves_per_s": rank_rates,
    }
    if DRY:
        out["dry_run"] = True
        out["data"] = "dry run: no device work (launch / reduction logic only)"
        print(json.dumps(out))
        return
    w0 = wins[0]
    pr = w0["prior"]
    bpi = algorithmic_bytes_per_iteration(len(w0["imu"]["frame_i"]), len(w0["proj"]["frame_i"]), len(w0["line"]["frame"]),
                                          len(w0["lam"]), pr["n"], sum(pr["sizes"]))
    k_ms = float(np.mean(solve_ms)) if solve_ms else None
    m_ms = float(np.mean(marg_ms)) if marg_ms else None
    # dominant kernel = the fused solve kernel.  SURVEY.md 8(d)'s unit of work is ONE trust-region iteration (B x 8 per launch, the same unit
    # as `iterations_per_s`): `achieved` / `frac` are quoted on it.  The kernel runs B x 9 linearisations for them (Ceres' iteration 0 is the
    # initial linearisation; each reads the window's algorithmic bytes): the per-linearisation figure stands beside it.
    units = B * SOLVER_ITERATIONS
    lins = B * (SOLVER_ITERATIONS + 1)
    achieved = (bpi * units) / (k_ms * 1e-3) / 1e9 if k_ms else None
    achieved_lin = (bpi * lins) / (k_ms * 1e-3) / 1e9 if k_ms else None
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")      # written by tools/pmc_traffic.py from a rocprofv3 --pmc run
    if os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile)).get("solve_kernel_hbm_bytes_per_launch")
        except (OSError, ValueError):
            traffic = None
    out["kernel_ms"] = {"solve": k_ms, "marginalize": m_ms}
    out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                       "kernel": "tcv::solve_kernel", "algorithmic_bytes_per_iteration": bpi,
                       "unit_of_work": "trust-region iteration (SURVEY.md 8(d)); same unit as iterations_per_s",
                       "units_per_launch": units, "linearisations_per_launch": lins,
                       "achieved_per_linearisation": achieved_lin, "frac_per_linearisation": (achieved_lin / HBM_PEAK_GBS) if achieved_lin else None,
                       "note": "fused FP64 solve: latency/issue bound, not HBM bound (DESIGN.md 4.1); frac is vs "
                               "the 8 TB/s HBM3E spec; traffic = PMC FETCH_SIZE x2 + WRITE_SIZE of profiles/pmc_traffic.json (static, from the committed profile run)"}
    # the same launches against the FP64 roof -- peak MEASURED on this device (tcv_microbench_fp64: dependence-free
    # v_fma_f64 / v_mfma_f64_16x16x4 chains on every CU), not quoted
    mb = (C.c_double * 4)()
    if tcv.lib().tcv_microbench_fp64(mb) == 0 and k_ms:
        peak = max(mb[0], mb[1])
        ach = FLOPS_PER_ITERATION_CFG3 * units / (k_ms * 1e-3) / 1e12
        out["roofline_fp64"] = {"bound": "fp64", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                                "peak_vector_fma": mb[0], "peak_mfma_f64_16x16x4": mb[1], "compute_units": int(mb[2]), "shader_clock_mhz": mb[3],
                                "flops_per_iteration": FLOPS_PER_ITERATION_CFG3, "units_per_launch": units,
                                "note": "peak = the better of the two micro-benchmarks run by this process on this device just now; flops = SURVEY.md 8(d)'s "
                                        "algorithmic 3.3 MFLOP per iteration and window (dense reduced-camera solve), not the instructions the fused kernel issues "
                                        "(those: `executed`, from the hardware counters of the committed profile run)"}
    # hardware counters of the same launch shape, from the committed profile run (tools/profile_round.sh): static, so they carry the commit
    # AND a hash of the kernel sources they were taken at -- when tc-viml_amd/csrc/ has changed since, they are not printed (`"stale": true`)
    cfile = os.path.join(ROOT, "profiles", "counters.json")
    if os.path.exists(cfile):
        try:
            cj = json.load(open(cfile))
            stale = cj.get("csrc_sha16") != csrc_sha16()
            if stale:
                out["counters"] = {"stale": True, "commit": cj.get("commit"), "csrc_sha16": cj.get("csrc_sha16"), "csrc_sha16_now": csrc_sha16(),
                                   "note": "profiles/counters.json was taken at other kernel sources than the ones this run was built from: figures withheld"}
                out["roofline"]["traffic"] = None
                out["roofline"]["traffic_stale"] = True
            else:
                out["counters"] = {k: cj.get(k) for k in ("commit", "csrc_sha16", "valu_active", "mfma_busy", "salu_per_valu", "waiting", "solve_kernel_hbm_bytes_per_launch",
                                                          "solve_kernel_write_bytes_per_launch", "marg_kernel_hbm_bytes_per_launch", "wait_split", "mean_latency_cycles",
                                                          "insts_per_launch", "l2_hit_rate", "source")}
                out["counters"]["stale"] = False
                out["mfma_busy"] = cj.get("mfma_busy"); out["valu_active"] = cj.get("valu_active")
                if traffic is None:
                    out["roofline"]["traffic"] = cj.get("solve_kernel_hbm_bytes_per_launch")
                out["roofline"]["traffic_commit"] = cj.get("commit")
                if "roofline_fp64" in out and cj.get("mfma_mops_f64_per_launch") and k_ms:
                    mf = cj["mfma_mops_f64_per_launch"] * 512.0      # one MOPS unit = 512 flops (SQ_INSTS_VALU_MFMA_MOPS_F64)
                    vu = (cj.get("valu_insts_per_launch") or 0.0) * 64 * 2
                    out["roofline_fp64"]["executed"] = {"mfma_flops_per_launch": mf, "mfma_tflops": mf / (k_ms * 1e-3) / 1e12, "mfma_frac_of_peak": mf / (k_ms * 1e-3) / 1e12 / mb[1] if mb[1] else None,
                                                        "valu_flops_per_launch_upper_bound": vu, "valu_tflops_upper_bound": vu / (k_ms * 1e-3) / 1e12,
                                                        "note": "counters of the committed profile run over THIS run's kernel time; the matrix-core figure counts the zero padding of the 9- and 15-wide "
                                                                "blocks in 16x16x4 tiles; the vector figure prices every VALU instruction as a 64-lane FMA (upper bound)"}
        except (OSError, ValueError):
            pass
    if world == 1 and not args.no_extras:
        out.update(sweep_figures(tcv, keep, opts))
    if world == 1 and not args.no_extras:
        # the PCIe-inclusive figure uses four host threads x 512 windows: the benchmark's 1024 windows and 1024 more of the same kind
        _b2, _w2, keep2 = build_batches(tcv, synth, shard_ids(rank, B) + B, B) if B == 1024 else (None, None, ([], [], []))
        del _b2
        out.update(stream_figures(tcv, torch, tuple(a + b for a, b in zip(keep, keep2)), wins=wins, local=local))
        out.update(replay_figures(tcv, local))
    if pool is not None:
        if "replay_windows_per_s" in out:      # the replay figure's own CPU baseline: the same eight streams through the oracle back end, one process each
            import replay
            rb = cpu_replay_baseline(pool, min(nproc, 8), replay.WINDOW_SIZE + 1 + 10 + 60, 60, 8, min(8.0, args.cpu_budget), close=False)      # (replay_figures' streams: 10 + 60 frames)
            if rb:
                out["replay_cpu_baseline"] = rb
                out["replay_over_cpu_all_cores"] = out["replay_windows_per_s"] / rb["value"]
                out["replay_over_cpu_core"] = out["replay_windows_per_s"] / rb["single_core_value"]
        out["cpu_baseline"] = cpu_baseline(wins, pool, nproc, args.cpu_budget)
        out["gpu_over_cpu_all_cores"] = out["value"] / out["cpu_baseline"]["value"]
        out["gpu_over_cpu_core"] = out["value"] / out["cpu_baseline"]["single_core_value"]
    print(json.dumps(out))


def run_stream(args, rank, world, local, dist):
    import torch
    import synth
    import tcv
    if tcv.lib().tcv_device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    tcv.check(tcv.lib().tcv_set_device(local))
    torch.cuda.set_device(local)
    _b, _w, keep = build_batches(tcv, synth, shard_ids(rank, args.windows), args.windows)
    out = stream_figures(tcv, torch, keep, rounds=max(1, args.steps // 4), wins=_w, local=local)
    if rank == 0:
        print(json.dumps(dict(metric="PCIe-inclusive streaming solves/sec (10 kf, 200 pt, 40 line)", value=out["stream_solves_per_s"], unit="solves/s",
                              n_gpus=1, higher_is_better=True, dtype="f64", data="synthetic", **out)))


class ReplayEngine:
    """EuRoC-trajectory streams through the native estimator (include/tcv_estimator.h), `groups` host threads each advancing its own
    share of the streams in lock step (one device batch per frame and group): while one group's kernels run, the others pack and
    upload.  Frames of ONE stream stay sequential (frame k + 1 needs frame k's states and prior).  A step = one frame of every stream."""

    def __init__(self, tcv, replay, stream_ids, n_frames, features, lines, groups, local, num_iterations=SOLVER_ITERATIONS, solver_time=0.0):
        self.tcv, self.local = tcv, local
        seqs = list(replay.EUROC_SEQUENCES)
        streams = [replay.simulate_stream_euroc(seqs[s % len(seqs)], n_frames, start_s=(0.5 + 3.0 * (s // len(seqs))) % max(3.0, 33.0 - 0.1 * n_frames), max_features=features,
                                                max_lines=lines, associate=True) for s in stream_ids]
        G = max(1, min(groups, len(streams)))
        # TCV_BENCH_PIPELINE=1 (experiment, NOT the default): every host thread drives TWO lock-step objects, each with half of the thread's
        # streams and its own library stream (slot 0 / 1), and alternates between them -- the host side of one half's frame against the kernels
        # of the other half's (tcv_estimators_optimize_begin / _end).  Measured slower than one object per thread at every stream count
        # (profiles/r05_replay_pipeline.txt): a group's own cycle -- its host side, then its solve -- is not shortened by halving the group, the
        # per-call fixed costs double, and four cooperative launches in flight crowd the chip
        self.pipelined = bool(os.environ.get("TCV_BENCH_PIPELINE")) and len(streams) >= 2 * G
        self.G = G
        H = 2 * G if self.pipelined else G
        self.ls = [replay.NativeLockstep(streams[h::H], num_iterations=num_iterations, solver_time=solver_time) for h in range(H)] if streams else []
        for h, ls in enumerate(self.ls):
            ls.slot = (h // G) if self.pipelined else 0
        for ls in self.ls:
            ls.prepare()
        self.k = 0
        while self.ls and self.k < n_frames:      # window fill: frames 0 .. WINDOW_SIZE, the last one triggers the first optimisation
            r = sum(ls.step(self.k) for ls in self.ls)
            self.k += 1
            if r:
                break
        if not self.ls:
            self.k = replay.WINDOW_SIZE + 1
        for ls in self.ls:      # the per-frame input records of the remaining frames (what a front end hands over), built ahead of the timed loop
            if self.k > replay.WINDOW_SIZE:
                ls.prepare_batches(self.k, n_frames)

    def run(self, steps: int) -> int:
        """`steps` frames of every stream; returns the number of windows optimised"""
        import threading
        k0, counts = self.k, [0] * len(self.ls)
        G = self.G

        def work(g):
            self.tcv.check(self.tcv.lib().tcv_set_device(self.local))      # the current device is per host thread
            if not self.pipelined:
                for k in range(k0, k0 + steps):
                    counts[g] += self.ls[g].step(k)
                return
            A, B = self.ls[g], self.ls[g + G]      # (slots 0 and 1 of this thread)
            if steps <= 0:
                return
            A.step_begin(k0)
            for k in range(k0, k0 + steps):
                B.step_begin(k)
                counts[g] += A.step_end()
                if k + 1 < k0 + steps:
                    A.step_begin(k + 1)
                counts[g + G] += B.step_end()

        # group 0 on the calling thread (its stream exists already), the others on threads of their own: one stream fewer on the runtime's
        # four hardware queues -- two host threads that share a queue wait for each other's solve kernels
        th = [threading.Thread(target=work, args=(g,)) for g in range(1, G)]
        for t in th:
            t.start()
        work(0)
        for t in th:
            t.join()
        self.k += steps
        return sum(counts)


def run_replay(args, rank, world, local, dist):
    """BASELINE configs[4]: EuRoC-trajectory streams sharded s mod G over the ranks; every rank advances its streams frame by frame."""
    mine = shard_streams(args.streams, rank, world)
    if args.host_threads <= 0:
        args.host_threads = 4 if len(mine) >= 96 else 2
    pool = nproc = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not DRY:
        pool, nproc = cpu_pool()                     # forked before the first GPU call
    if DRY:
        dev = None
        run = lambda steps: (time.sleep(0.001 * steps), len(mine) * steps)[1]
    else:
        import torch
        import replay
        import tcv
        if tcv.lib().tcv_device_count() < 1:
            raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
        tcv.check(tcv.lib().tcv_set_device(local))
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        eng = ReplayEngine(tcv, replay, mine, replay.WINDOW_SIZE + 1 + args.warmup + args.steps, args.features, args.lines, args.host_threads, local)
        run = eng.run
    red_dev = dev if (dist is not None and dist.get_backend() == "nccl") else None
    n_ranks = ranks_seen(dist, red_dev)
    run(args.warmup)
    if not DRY:
        import ctypes as C
        tcv.lib().tcv_estimators_profile((C.c_double * 8)())      # clears the accounting of the window fill and the warm-up
        tcv.lib().tcv_estimators_kernel_profile((C.c_double * 8)())
        for ls in eng.ls:
            ls.host_s = [0.0, 0.0, 0.0, 0]
    if dist is not None:
        dist.barrier()
    dbg0 = None
    if not DRY and os.environ.get("TCV_BENCH_DEBUG_POOLS"):
        Lb = tcv.lib(); Lb.tcv_debug_host_pool_allocs.restype = C.c_longlong; Lb.tcv_debug_dev_pool_misses.restype = C.c_longlong
        dbg0 = (Lb.tcv_debug_host_pool_allocs(), Lb.tcv_debug_dev_pool_misses())
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    n = run(args.steps)
    cpu_s = time.process_time() - cpu0               # every thread of this process: host threads, packer workers, HIP runtime threads
    if dbg0 is not None:
        a, b = Lb.tcv_debug_host_pool_allocs(), Lb.tcv_debug_dev_pool_misses()
        print("[bench] timed region: hipHostMalloc %d, hipHostFree %d, hipMalloc %d" % (a // 1000000 - dbg0[0] // 1000000, a % 1000000 - dbg0[0] % 1000000, b - dbg0[1]), file=sys.stderr)
    if not DRY:
        import torch
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed_max, windows_total, iters_total = reduce_stats(dist, elapsed, n, red_dev, extra=(n * SOLVER_ITERATIONS,))
    rank_rates = gather_rank_rates(dist, elapsed, n, red_dev)
    if rank != 0:
        return
    prof = None
    if not DRY:      # host-side accounting of tcv_estimators_optimize on rank 0 (summed over its host threads), per call
        import ctypes as C
        p8 = (C.c_double * 8)()
        tcv.lib().tcv_estimators_profile(p8)
        names = ["preintegrate", "assoc+triangulate+window", "problems", "batch_create", "kernels", "downloads", "apply"]
        prof = {k: round(1e3 * v / max(1.0, p8[7]), 3) for k, v in zip(names, p8)}
        prof["calls"] = int(p8[7])
        # around the native call, per call: tcv_estimator_begin_frame (IMU propagation, feature bookkeeping, keyframe test) and
        # tcv_estimator_get_stats / finish_frame (failure detection, window slide) of the call's streams, Python harness included
        hs = [sum(ls.host_s[i] for ls in eng.ls) for i in range(4)]
        prof["begin_frames"] = round(1e3 * hs[0] / max(1, hs[3]), 3); prof["finish_frames"] = round(1e3 * hs[2] / max(1, hs[3]), 3)
    out = {"metric": "sliding-window solves/sec, EuRoC-trajectory replay (configs[4])", "value": windows_total / elapsed_max, "unit": "solves/s",
           "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed_max / args.steps, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f64",
           "data": "synthetic front-end streams along the EuRoC ground-truth trajectories the reference ships (no bag in the tree)",
           "config": {"workload": f"configs[4]: {args.streams} EuRoC-trajectory streams (V1_02..V2_03 excerpts, {args.features} tracked features + {args.lines} line "
                                  f"tracks per frame, 2D-3D association in the loop), stream s on rank s mod {world}, native estimator, "
                                  f"{SOLVER_ITERATIONS} iterations (convergence tests on) + marginalisation per frame; per rank {args.host_threads} host threads, "
                                  f"each advancing its share of the rank's streams in lock step (cooperative small-batch kernels)",
                      "streams": args.streams, "host_threads": args.host_threads, "parallelism": f"streams sharded x{world}",
                      "hip_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")},
           "frames_per_s_per_stream": windows_total / elapsed_max / max(1, args.streams),
           "ranks_seen": n_ranks, "per_rank_solves_per_s": rank_rates, "streams_per_rank": [len(shard_streams(args.streams, r, world)) for r in range(world)]}
    if DRY:
        out["dry_run"] = True
    out["host_cpu"] = {"cores_busy_mean": cpu_s / max(elapsed, 1e-9), "cpu_us_per_window": 1e6 * cpu_s / max(1, n), "cgroup_cpu_quota_cores": _cpu_quota(),
                       "note": "rank 0's process CPU time over the timed frames / their wall time: how much of the granted cores the host side of the replay keeps busy"}
    if prof:
        out["native_profile_ms_per_call"] = prof
    if not DRY:
        # every replay window has a structure of its own: the whole-plan cache misses, the camera half of the plan (IMU tables, chain records,
        # prior tables -- tcv_pack.cpp) is found; the visual half is built per window and frame
        pc = (C.c_longlong * 4)()
        if tcv.lib().tcv_plan_cache_stats(pc) == 0:
            out["plan_cache"] = {"whole_plan_hits": int(pc[0]), "whole_plan_misses": int(pc[1]), "camera_half_hits": int(pc[2]), "camera_half_misses": int(pc[3]),
                                 "camera_half_hit_rate": pc[2] / max(1, pc[2] + pc[3]), "note": "since the process started (window fill and warm-up included)"}
        # the kernels this path runs, live: HIP-event durations of every solve / marginalisation launch of the timed frames (rank 0) and the
        # ALGORITHMIC bytes of the windows they held (SURVEY.md 8(d) formula on every window's own factor counts x its linearisations)
        k8 = (C.c_double * 8)()
        tcv.lib().tcv_estimators_kernel_profile(k8)
        if k8[0] > 0 and k8[1] > 0:
            ach = k8[5] / (k8[0] * 1e-3) / 1e9
            out["kernel_ms"] = {"solve": k8[0] / k8[1], "solve_launches": int(k8[1]), "windows_per_solve_launch": k8[4] / k8[1],
                                "marginalize": (k8[2] / k8[3]) if k8[3] > 0 else None, "marginalize_launches": int(k8[3]),
                                "windows_per_marginalize_launch": (k8[7] / k8[3]) if k8[3] > 0 else None}
            out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                               "kernel": "tcv::solve_kernel<256, true, true, true, false> (cooperative: 1 + H workgroups per window)",
                               "algorithmic_bytes_per_linearisation_mean": k8[5] / max(1.0, k8[6]), "linearisations_per_window_mean": k8[6] / max(1.0, k8[4]),
                               "note": "achieved = algorithmic bytes of the windows of all launches (each window's own factor counts, x its linearisations) / summed HIP-event "
                                       "duration of the launches; launches of different host threads overlap on the device, so the sum of durations exceeds the wall time "
                                       "they occupy.  A lock-step frame of a few windows is LATENCY bound (a window's trust-region iterations are sequential; the chip is "
                                       "mostly idle): the fraction says how far from a bandwidth problem this workload is, not how good the kernel is"}
    if pool is not None:
        # (the generator is called with the GPU run's own frame count: identical streams)
        try:
            cb = cpu_replay_baseline(pool, min(nproc, max(1, args.streams)), replay.WINDOW_SIZE + 1 + args.warmup + args.steps, args.features, args.lines, args.cpu_budget)
        except Exception as e:      # the GPU line stands on its own
            cb = None
            out["cpu_baseline_error"] = repr(e)[:300]
        if cb:
            out["cpu_baseline"] = cb
            out["gpu_over_cpu_all_cores"] = out["value"] / cb["value"]
            out["gpu_over_cpu_core"] = out["value"] / cb["single_core_value"]
    print(json.dumps(out))


def replay_figures(tcv, local, streams=8, groups=2, steps=60, warmup=10):
    """the replay number of the default line (one GPU): optimised windows per second of `streams` EuRoC-trajectory streams"""
    import replay
    eng = ReplayEngine(tcv, replay, list(range(streams)), replay.WINDOW_SIZE + 1 + warmup + steps, 60, 8, groups, local)
    import torch
    eng.run(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = eng.run(steps)
    torch.cuda.synchronize()      # (the last frame's marginalisation is launched behind the states the call returns with)
    dt = time.perf_counter() - t0
    out = {"replay_windows_per_s": n / dt,
           "replay_note": f"{streams} EuRoC-trajectory streams (60 features + 8 line tracks per frame, association in the loop) through the native estimator on "
                          f"this GPU, {groups} host threads x {streams // max(1, groups)} streams in lock step, {steps} frames per stream: {1e3 * dt / steps:.2f} ms per frame of every stream "
                          f"(bench.py --mode replay prints this as its value)"}
    del eng
    try:      # (an extra: whatever goes wrong in it must not cost the headline line)
        out["deployed_budget"] = deployed_budget_figures(tcv, replay, local, streams, groups, steps=max(20, steps // 2), warmup=warmup)
    except Exception as e:      # noqa: BLE001
        out["deployed_budget"] = {"error": repr(e)[:300]}
    return out


def deployed_budget_figures(tcv, replay, local, streams=8, groups=2, steps=30, warmup=10):
    """The same replay on the solver budget the reference SHIPS (benchmark_publisher/config/V1_01_easy/sensor.yaml:85-86, estimator.cpp:1892-1897):
    max_num_iterations 100, convergence tests on, max_solver_time 0.04 s (x 4/5 on frames that marginalise the oldest keyframe) -- every other
    replay figure of this file uses 8 iterations.  Reports how the solves ended: a GPU window that needs its ~26 iterations is done long before
    the 32 ms wall budget that cuts the reference's CPU solve short."""
    import torch
    eng = ReplayEngine(tcv, replay, list(range(streams)), replay.WINDOW_SIZE + 1 + warmup + steps, 60, 8, groups, local, num_iterations=100, solver_time=0.04)
    eng.run(warmup)
    torch.cuda.synchronize()
    for ls in eng.ls:      # (only the timed frames are counted below)
        ls._flush_raw()
        for o in ls.outs:
            o["log"].clear()
    t0 = time.perf_counter()
    n = eng.run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    its, term = [], []
    for ls in eng.ls:
        for r in ls.results():
            its += [e["iterations"] for e in r["log"]]; term += [e["termination"] for e in r["log"]]
    its = np.array(its if its else [0]); term = np.array(term if term else [0])
    names = {0: "no_convergence_cap_or_clock", 1: "gradient_tolerance", 2: "parameter_tolerance", 3: "function_tolerance", 4: "radius", 5: "failure"}
    return {"windows_per_s": n / dt, "ms_per_frame": 1e3 * dt / steps, "max_num_iterations": 100, "max_solver_time_in_seconds": 0.04,
            "max_solver_time_margin_old": 0.032, "iterations_mean": float(its.mean()), "iterations_max": int(its.max()),
            "terminations": {names.get(int(k), str(int(k))): int((term == k).sum()) for k in np.unique(term)}, "windows": int(len(its)),
            "note": f"{streams} streams / {groups} host threads as replay_windows_per_s; iterations = summary.iterations.size() (iteration 0 included); a lock-step frame "
                    "is one batch with one budget: 0.032 s as soon as one of its windows is MARGIN_OLD (include/tcv_estimator.h)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=("solve", "replay", "stream"), default="solve")
    ap.add_argument("--windows", type=int, default=1024, help="independent windows per GPU per step")
    ap.add_argument("--threads", type=int, default=256)
    ap.add_argument("--variant", type=int, default=0, help="0: chain layout (default), 1: dense 171-dim layout (cross-check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the PCIe-inclusive / single-window figures")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--streams", type=int, default=8, help="replay mode: number of EuRoC-trajectory streams of the whole job")
    ap.add_argument("--features", type=int, default=60)
    ap.add_argument("--lines", type=int, default=8)
    ap.add_argument("--host-threads", type=int, default=0, help="replay mode: host threads per rank, each advancing its share of the rank's streams "
                    "(default: 2, and 4 from 96 streams per rank on -- profiles/r05_replay_host_workers.txt)")
    args = ap.parse_args()
    if args.mode == "stream" and args.gpus > 1:
        raise SystemExit("--mode stream measures one GPU (host threads x HIP streams of one device): use --gpus 1")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))       # nothing above touched the GPU
    rank, world, local, dist = dist_setup(args.gpus)
    try:
        {"solve": run_solve, "replay": run_replay, "stream": run_stream}[args.mode](args, rank, world, local, dist)
    finally:
        if dist is not None:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
