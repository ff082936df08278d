#!/usr/bin/env python3
"""Developer checker (GPU): the NATIVE estimator (include/tcv_estimator.h) on seeded stress streams -- 4 .. 120 tracked features, 0 .. 20 line
tracks, pixel noise 0.3 .. 3, slow and fast pacing, the five EuRoC trajectories from random start times, with / without the association in the
loop, ESTIMATE_EXTRINSIC on / off.  Every window the estimator optimises (its own states, its device-resident pre-integrations, the device-made
prior of its previous frame: tcv_estimator_set_window_tap) is solved again by the C oracle: iteration count, final cost, gauge-fixed states.
Windows of a starved front end (4 features) are ill posed; they are priced by the oracle's own movement under 1e-13 input noise like in
fuzz_solve.py.

    python tests/dev/fuzz_estimator.py [streams] [first seed] [frames]
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "tc-viml_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import np_oracle as NO      # noqa: E402
import orc      # noqa: E402
import replay      # noqa: E402
import tcv      # noqa: E402
from replay_cache import MODES      # noqa: E402
from test_gpu_teacher import _Snapshot, _snapshot_window      # noqa: E402
from util import rel      # noqa: E402


def oracle_window(win, extrinsic):
    O = orc.Window(win, ex_constant=not extrinsic)
    so = O.solve(8, False)
    sto = O.states()
    R0 = NO.q2R(win["pose"][0, 3:]); P0 = win["pose"][0, :3]
    Rs, Ps, Vs, po = orc.gauge_fix(R0, P0, sto["pose"], sto["sb"])
    sbo = sto["sb"].copy(); sbo[:, :3] = Vs
    return so, dict(pose=po, sb=sbo, lam=sto["lam"], ex=sto["ex"])


def main():
    streams = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    L = tcv.lib()
    L.tcv_estimator_set_window_tap.argtypes = [C.c_void_p, C.c_int]
    L.tcv_estimator_get_window_snapshot.argtypes = [C.c_void_p, C.POINTER(_Snapshot)]
    tot = dict(windows=0, ok=0, soft=0, iters=0, bad=0, not_applied=0)
    flagged = []
    for s in range(streams):
        seed = seed0 + s
        rng = np.random.Generator(np.random.PCG64(seed))
        seq = str(rng.choice(replay.EUROC_SEQUENCES)); mode = str(rng.choice(["associate", "given", "none"])); extrinsic = bool(rng.random() < 0.7)
        nf = int(rng.choice([4, 10, 25, 60, 120])); sig = float(rng.choice([0.3, 1.0, 3.0])); start = float(rng.uniform(0.2, 20.0))
        kw = dict(MODES[mode]); kw["max_features"] = nf; kw["pixel_sigma"] = sig
        if "max_lines" in kw and mode != "none":
            kw["max_lines"] = int(rng.choice([1, 4, 20]))
        note = f"{seq} from {start:.1f} s, {mode}, {nf} features, sigma {sig}, lines {kw.get('max_lines')}, extrinsic {'free' if extrinsic else 'constant'}"
        try:
            st = replay.simulate_stream_euroc(seq, frames, start_s=start, seed=seed, **kw)
        except ValueError as e:
            print(f"stream {seed} [{note}]: {e}"); continue
        ls = replay.NativeLockstep([st], num_iterations=8, estimate_extrinsic=extrinsic)
        tcv.check(L.tcv_estimator_set_window_tap(ls.ests[0], 1))
        row = dict(windows=0, ok=0, soft=0, iters=0, bad=0, not_applied=0)
        worst = 0.0
        try:
            for k in range(ls.n_frames):
                if not ls.step(k):
                    continue
                S = _Snapshot()
                tcv.check(L.tcv_estimator_get_window_snapshot(ls.ests[0], C.byref(S)))
                win, res = _snapshot_window(tcv, S)
                row["windows"] += 1
                if res["applied"] != 1:
                    row["not_applied"] += 1; continue
                so, ref = oracle_window(win, extrinsic)
                if so.num_iterations != res["iterations"]:
                    row["iters"] += 1; continue
                d = dict(cost=abs(res["cost"] - so.final_cost) / max(abs(so.final_cost), 1e-12), pose=rel(res["pose"], ref["pose"]), sb=rel(res["sb"], ref["sb"]),
                         lam=rel(res["lam"], ref["lam"]) if len(ref["lam"]) else 0.0, ex=rel(res["ex"], ref["ex"]))
                worst = max(worst, max(d.values()))
                if all(v < 1e-6 for v in d.values()):
                    row["ok"] += 1; continue
                sens = dict(cost=0.0, pose=0.0, sb=0.0, lam=0.0, ex=0.0)      # the oracle's own movement under 1e-13 input noise
                for rep in range(2):
                    r2 = np.random.Generator(np.random.PCG64(977 + rep))
                    w2 = dict(win)
                    for key in ("lam", "pose", "speedbias"):
                        a = np.asarray(win[key], dtype=float)
                        w2[key] = a * (1 + 1e-13 * r2.standard_normal(a.shape))
                    so2, ref2 = oracle_window(w2, extrinsic)
                    sens["cost"] = max(sens["cost"], abs(so2.final_cost - so.final_cost) / max(abs(so.final_cost), 1e-12))
                    for key in ("pose", "sb", "lam", "ex"):
                        sens[key] = max(sens[key], rel(ref2[key], ref[key]) if len(ref[key]) else 0.0)
                if all(d[key] < max(1e-6, 30 * sens[key]) for key in d):
                    row["soft"] += 1
                else:
                    row["bad"] += 1
                    flagged.append((seed, k, {key: f"{d[key]:.1e}/{sens[key]:.1e}" for key in d if d[key] >= 1e-6}))
        finally:
            ls.close()
        for key in tot:
            tot[key] += row[key]
        print(f"stream {seed} [{note}]: {row}, worst {worst:.1e}", flush=True)
    print("\ntotal:", tot)
    print("flagged windows (device difference / oracle's own movement):", len(flagged))
    for x in flagged[:20]:
        print("  ", x)
    return 1 if flagged else 0


if __name__ == "__main__":
    sys.exit(main())
