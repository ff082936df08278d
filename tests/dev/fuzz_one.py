#!/usr/bin/env python3
"""Developer checker (GPU): one case of tests/dev/fuzz_solve.py in detail -- where the first step and the states differ from the oracle, block by block.
    python tests/dev/fuzz_one.py <seed> [gone indices ...]      (the IMU factors to leave out override the case's own)"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fuzz_solve as fz
from util import fro, rel

seed = int(sys.argv[1])
rng = np.random.Generator(np.random.PCG64(seed))
w_hip, w_orc, exc, note = fz.make_case(rng, seed)
if len(sys.argv) > 2:
    gone = [int(a) for a in sys.argv[2:] if int(a) >= 0]
    base = dict(w_orc)      # (the case's own omissions are already applied to w_orc: rebuild from the hip window)
    im = dict(w_hip["imu"]); n = len(im["sum_dt"])
    sd = np.array(im["sum_dt"], float); sd[sd > 10] = 0.05; sd[gone] = 11.0; im["sum_dt"] = sd
    w_hip = dict(w_hip, imu=im)
    keep = ~np.isin(np.arange(n), gone)
    im0 = dict(w_hip["imu"]); sd0 = np.array(im0["sum_dt"], float); sd0[sd0 > 10] = 0.05; im0["sum_dt"] = sd0
    w_orc = dict(w_hip, imu=fz.take(im0, keep, n))
print(note, "| point factors", len(w_hip["proj"]["landmark"]), "lines", len(w_hip["line"]["frame"]), "landmarks", len(w_hip["lam"]), "frames", len(w_hip["pose"]))
print("imu pairs (oracle):", list(zip(w_orc["imu"]["frame_i"], w_orc["imu"]["frame_j"])))
pr = w_hip["proj"]
print("point factor frames:", sorted(set(zip([int(a) for a in pr["frame_i"]], [int(a) for a in pr["frame_j"]]))))
O = fz.orc.Window(w_orc, ex_constant=exc); so = O.solve(8, True)
fo = np.array(so.first_delta[:so.n_local])
sys.path.insert(0, os.path.join(fz.ROOT, "tests", "golden"))
try:
    import make_golden_pins as mgp
    mp = None if exc else np.asarray(mgp.mp_first_step(w_orc)["delta"], dtype=float)
except Exception as e:      # noqa: BLE001
    print("mpmath first step unavailable:", e); mp = None
if mp is not None:
    print("oracle first step vs the 50-digit solution of the same regularised system:", fro(fo, mp))
for name, kw in (("lone", dict(copies=1)), ("lone, substitution instead of explicit inverses (use_mfma = 0)", dict(copies=1, mfma=False)), ("dense", dict(copies=1, dense=True)), ("dense, use_mfma = 0", dict(copies=1, dense=True, mfma=False))):
    Ws, b, s = fz.gpu_run(w_hip, exc, **kw)
    fg = b.first_step(0)
    print(name, "layout", b.plan_stats()["layout"], "first step rel", fro(fg, fo), "n", len(fg), len(fo), "" if mp is None else "| vs 50-digit solution %.2e" % fro(fg, mp))
    F = len(w_hip["pose"])
    d = np.abs(fg - fo)
    # local parameter order of the oracle: poses (6 each), speed-bias (9 each), extrinsic (6), landmarks -- print the worst entries
    idx = np.argsort(-d)[:12]
    print("   worst entries (index, gpu, oracle):", [(int(i), float(fg[i]), float(fo[i])) for i in idx])
    print("   costs gpu", [s[0].cost[i] for i in range(s[0].num_iterations)], "oracle", [so.cost[i] for i in range(so.num_iterations)])
    print("   mu / radius trace: dogleg cases gpu", [s[0].dogleg_case[i] for i in range(1, s[0].num_iterations)], "oracle", [so.dogleg_case[i] for i in range(1, so.num_iterations)])
