"""BASELINE configs[3] / configs[4]: replay along the EuRoC ground-truth trajectories the reference ships (five sequences,
36 s excerpts, tc-viml_amd/data/euroc_*.npz) through the HIP back end; the sequences run in lock step, one device batch per frame.

    python tools/replay_euroc.py [--frames 340] [--line-mode associate|given|none] [--out gpurun_out/euroc]
    python -m torch.distributed.run --nproc-per-node N ... tools/replay_euroc.py     # sequences sharded rank r -> r::N (configs[4])

Per sequence: `vins_result_<seq>.csv` in the reference's format (visualization.cpp:211-226), the ground truth rows in the format
benchmark_publisher parses, and the ATE (SE(3)-aligned RMSE) of the estimate against the ground truth.  The bag (images, raw IMU)
is not part of the reference: the front-end streams are simulated on the trajectory (replay.simulate_stream_euroc); the paper's
Table II numbers are for the real images and are quoted for orientation only."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd"))
import numpy as np
import replay, ate

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=340)
ap.add_argument("--start", type=float, default=0.5)
ap.add_argument("--features", type=int, default=60)
ap.add_argument("--lines", type=int, default=8)
ap.add_argument("--line-mode", choices=("associate", "given", "none"), default="associate",
                help="associate: the 2D-3D association (tcv_match_lines + removeLineOutlier) runs in the loop, as in the reference; "
                     "given: every line observation arrives with its true 3D partner; none: no line factors")
ap.add_argument("--exact-line-jacobian", action="store_true", help="opt-in extension: derivative of the line residual instead of the reference's Jacobian (tcv_problem_set_line_jacobian)")
ap.add_argument("--native", action="store_true", help="window management in native code (include/tcv_estimator.h) instead of replay.Replay")
ap.add_argument("--profile", action="store_true", help="print the host-side time accounting of the native estimator (tcv_estimators_profile)")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "euroc"))
args = ap.parse_args()
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    import torch
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    import tcv
    tcv.check(tcv.lib().tcv_set_device(int(os.environ.get("LOCAL_RANK", "0"))))
seqs = list(replay.EUROC_SEQUENCES)[rank::world]
os.makedirs(args.out, exist_ok=True)
t0 = time.perf_counter()
streams = [replay.simulate_stream_euroc(s, args.frames, start_s=args.start, max_features=args.features, max_lines=0 if args.line_mode == "none" else args.lines,
                                         associate=args.line_mode == "associate") for s in seqs]
t1 = time.perf_counter()
outs = (replay.run_many_native(streams, num_iterations=8, exact_line_jacobian=args.exact_line_jacobian) if args.native else replay.run_many(streams, replay.HipBackend(), num_iterations=8, exact_line_jacobian=args.exact_line_jacobian)) if streams else []
t2 = time.perf_counter()
rows = []
for st, o in zip(streams, outs):
    stamps = st["stamp_ns"][np.searchsorted(st["t"], o["t"])] * 1e-9
    ate.write_vins_result(os.path.join(args.out, "vins_result_%s.csv" % st["seq"]), stamps, o["p"], o["q"], o["v"])
    i, j = ate.associate(o["t"], st["t"])
    flags = [l["flag"] for l in o["log"]]
    rows.append(dict(seq=st["seq"], optimised_frames=len(o["t"]), seconds=round(float(o["t"][-1] - o["t"][0]), 1),
                     path_length_m=round(float(np.linalg.norm(np.diff(st["gt_p"], axis=0), axis=1).sum()), 1),
                     ate_aligned_m=round(ate.ate_rmse(o["p"][i], st["gt_p"][j]), 4), ate_raw_m=round(ate.ate_rmse(o["p"][i], st["gt_p"][j], align=False), 4),
                     margin_old=flags.count(replay.MARGIN_OLD), margin_second_new=flags.count(replay.MARGIN_SECOND_NEW),
                     point_factors_mean=round(float(np.mean([l["n_proj"] for l in o["log"]])), 1), line_factors_mean=round(float(np.mean([l["n_line"] for l in o["log"]])), 1)))
frames = sum(r["optimised_frames"] for r in rows)
res = dict(rank=rank, world=world, line_mode=args.line_mode, line_jacobian="exact" if args.exact_line_jacobian else "reference", window_management="native" if args.native else "python", sequences=rows, optimised_frames=frames, simulate_s=round(t1 - t0, 2), replay_s=round(t2 - t1, 2),
           frames_per_s=round(frames / max(t2 - t1, 1e-9), 1))
if args.native and args.profile:
    import ctypes as C, tcv
    prof = (C.c_double * 8)()
    tcv.lib().tcv_estimators_profile(prof)
    names = ["preintegrate", "assoc+triangulate+window", "problems", "batch_create", "kernels", "downloads", "apply", "calls"]
    res["native_profile_s"] = {k: round(v, 4) for k, v in zip(names, prof)}
with open(os.path.join(args.out, "replay_euroc_%s%s_rank%d.json" % (args.line_mode, "_exactJ" if args.exact_line_jacobian else "", rank)), "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res))
