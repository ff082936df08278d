#!/usr/bin/env python3
"""Headline benchmark: sliding-window solves/sec on the BASELINE.json workload
("10 kf, 200 pt, 40 line" + marginalisation prior = configs[2]) on N MI355X GPUs of one node.

One STEP = one pass of the hot path over one batch of B independent synthetic windows per GPU, already
resident in HBM: the fused solve kernel (8 fixed trust-region iterations, the per-frame budget regime of
estimator.cpp:1888-1897 made deterministic), the double2vector gauge fix (:1905) and the MARGIN_OLD marginalisation kernel
(estimator.cpp:1911-2046) that produces the next prior.  value = windows solved per second over all GPUs.

Multi-GPU: windows are independent (per-sequence replay shards one sequence per GPU), so every rank owns
B windows with rank-offset seeds and there is no data-path collective; RCCL is used only for the barrier
and the MAX-over-ranks time ("scaling": "weak").

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        bench.py --gpus 8 --steps 20 --warmup 3
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd"))

HBM_PEAK_GBS = 8000.0                 # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
SOLVER_ITERATIONS = 8
# ALGORITHMIC bytes per trust-region iteration per window (cfg 3: 200 pt, 40 line, prior n=75, L=50)
BYTES_PER_ITERATION_CFG3 = 89056


def algorithmic_bytes_per_iteration(n_imu, n_pt, n_ln, L, prior_n, prior_x0):
    """8*[S + N_imu*287 + N_pt*6 + N_ln*9 + 21 + (n^2 + n + X0)] + 4*[4N_imu + 4N_pt + N_ln] + 8*[(171+L) + 1]."""
    S = 77 + 99 + 7 + L
    prior = (prior_n * prior_n + prior_n + prior_x0) if prior_n else 0
    return 8 * (S + n_imu * 287 + n_pt * 6 + n_ln * 9 + 21 + prior) + 4 * (4 * n_imu + 4 * n_pt + n_ln) + 8 * ((171 + L) + 1)


def shard_ids(rank: int, per_gpu: int) -> int:
    """first synthetic window id of a rank: disjoint seeds per rank, fixed work per GPU (weak scaling)."""
    return 100000 + rank * per_gpu


def dist_setup(n_gpus: int, backend: str | None = None):
    """(rank, world, local_rank, dist-or-None).  For N > 1 every rank is one process launched by torch.distributed.run."""
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return 0, 1, 0, None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local, dist


def reduce_stats(dist, elapsed_s: float, windows: int, device=None):
    """MAX of the elapsed time and SUM of the windows over ranks (the only collectives of the whole job)."""
    if dist is None:
        return elapsed_s, windows
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    w = torch.tensor([float(windows)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(w, op=dist.ReduceOp.SUM)
    return float(t.item()), int(w.item())


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
    b0.solve(opts); b0.marginalize(); b0.synchronize()
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
    return batch, wins, (Wm, Mm, dropsm)


def cpu_baseline(wins, budget_s: float = 12.0):
    """the CPU restatement (oracle/tcv_oracle.c, one thread like Ceres num_threads = 1) on a bounded sample of
    the SAME windows: solve (8 fixed iterations) + MARGIN_OLD marginalisation per window."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    orc.lib()
    n, t0 = 0, time.perf_counter()
    ws = [orc.Window(w) for w in wins[:min(len(wins), 4096)]]
    t0 = time.perf_counter()
    for O in ws:
        O.solve(SOLVER_ITERATIONS, True)
        O.marginalize_old()
        n += 1
        if time.perf_counter() - t0 > budget_s and n >= 16:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "solves/s", "cores": 1, "kind": "port",
            "sample": f"{n} of the benchmark's windows, {SOLVER_ITERATIONS} fixed iterations + 1 marginalisation each, "
                      f"{dt:.1f} s on one host core (oracle/tcv_oracle.c, gcc -O3; dense Schur, not Ceres)",
            "host_cpu": _cpu_model(), "host_cores_available": os.cpu_count()}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=1024, help="independent windows per GPU per step")
    ap.add_argument("--threads", type=int, default=256)
    ap.add_argument("--variant", type=int, default=0, help="0: chain layout (default, two windows per CU), 1: dense 171-dim layout (cross-check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()

    rank, world, local, dist = dist_setup(args.gpus)
    import torch
    import synth
    import tcv
    if tcv.lib().tcv_device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    tcv.check(tcv.lib().tcv_set_device(local))
    tcv.check(tcv.lib().tcv_set_solver_variant(args.variant))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B = args.windows
    batch, wins, _keep = build_batches(tcv, synth, shard_ids(rank, B), B)
    opts = tcv.default_options(SOLVER_ITERATIONS, True, True, args.threads)

    def step():
        batch.solve(opts)
        batch.gauge_fix()          # double2vector(), estimator.cpp:1905: the marginalisation linearises at the gauge-fixed states
        batch.marginalize()

    def sync():
        batch.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync()
    solve_ms, marg_ms = [], []
    t0 = time.perf_counter()
    sample_every = max(1, (args.steps + 63) // 64)      # HIP-event duration of the launches (events recorded on the launch stream): at most 64 samples
    for k in range(args.steps):
        step()
        if k % sample_every == 0:
            batch.synchronize()
            st = batch.stats(); solve_ms.append(st["solve_ms"]); marg_ms.append(st["marg_ms"])
    sync()
    elapsed = time.perf_counter() - t0
    elapsed_max, windows_total = reduce_stats(dist, elapsed, B * args.steps, dev if dist is not None and dist.get_backend() == "nccl" else None)

    # parity spot-check of what was just timed is done by tests/ and smoke(); here only sanity of the results
    s = batch.summaries(min(4, B))
    assert all(np.isfinite(s[k].final_cost) and s[k].final_cost < s[k].initial_cost for k in range(min(4, B)))

    if rank == 0:
        w0 = wins[0]
        pr = w0["prior"]
        bpi = algorithmic_bytes_per_iteration(len(w0["imu"]["frame_i"]), len(w0["proj"]["frame_i"]), len(w0["line"]["frame"]),
                                              len(w0["lam"]), pr["n"], sum(pr["sizes"]))
        k_ms = float(np.mean(solve_ms)) if solve_ms else None
        m_ms = float(np.mean(marg_ms)) if marg_ms else None
        # dominant kernel = the fused solve kernel: B windows x (initial linearisation + 8 iterations) per launch
        units = B * (SOLVER_ITERATIONS + 1)
        achieved = (bpi * units) / (k_ms * 1e-3) / 1e9 if k_ms else None
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")      # written by tools/pmc_traffic.py from a rocprofv3 --pmc run
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get("solve_kernel_hbm_bytes_per_launch")
            except (OSError, ValueError):
                traffic = None
        out = {
            "metric": "sliding-window solves/sec (10 kf, 200 pt, 40 line)",
            "value": windows_total / elapsed_max,
            "unit": "solves/s",
            "n_gpus": world,
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
            "iterations_per_s": windows_total * SOLVER_ITERATIONS / elapsed_max,
            "kernel_ms": {"solve": k_ms, "marginalize": m_ms},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "kernel": "tcv::solve_kernel", "algorithmic_bytes_per_iteration": bpi,
                         "units_per_launch": units,
                         "note": "fused FP64 solve (chain layout, two windows per CU): latency/issue bound, not HBM bound (DESIGN.md 4.1); frac is vs "
                                 "the 8 TB/s HBM3E spec; traffic = PMC FETCH_SIZE x2 + WRITE_SIZE of profiles/pmc_traffic.json"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(wins, args.cpu_budget)
            out["gpu_over_cpu_core"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
