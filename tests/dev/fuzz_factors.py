#!/usr/bin/env python3
"""Developer checker (GPU): the residual / Jacobian evaluators (ProjectionFactor, LineProjectionFactor, IMUFactor: the C-ABI's
CostFunction::Evaluate layout) on seeded random inputs far outside the synthetic windows -- arbitrary attitudes with non-unit quaternions,
inverse depths from 1e-3 to 1e3 and negative, points near the image plane's horizon, poses metres apart, IMU intervals up to 10 s with large
bias corrections -- against the NumPy oracle.  Gate: 1e-10 of the largest entry of each quantity (1e-9 for the IMU factor), as in
tests/test_gpu_factors.py.

    python tests/dev/fuzz_factors.py [cases per factor] [first seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import np_oracle as npo      # noqa: E402
import synth      # noqa: E402
import tcv      # noqa: E402
from util import rel      # noqa: E402

NOISE = (synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W)


def rand_pose(rng, spread, unit=True):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    if not unit:
        q *= 1.0 + float(rng.choice([1e-9, 1e-3, 5e-2])) * rng.normal()
    return np.concatenate([rng.normal(size=3) * spread, q])


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(300000 + seed0)
    bad = 0
    # ---- P1
    pi = np.array([rand_pose(rng, float(rng.choice([0.1, 2.0, 30.0])), unit=rng.random() < 0.6) for _ in range(n)])
    pj = np.array([rand_pose(rng, float(rng.choice([0.1, 2.0, 30.0])), unit=rng.random() < 0.6) for _ in range(n)])
    ex = np.array([rand_pose(rng, 0.2, unit=rng.random() < 0.6) for _ in range(n)])
    lam = np.exp(rng.uniform(np.log(1e-3), np.log(1e3), n)) * rng.choice([1.0, 1.0, 1.0, -1.0], n)
    pti = np.concatenate([rng.normal(size=(n, 2)) * rng.choice([0.1, 1.0, 10.0], (n, 1)), np.ones((n, 1))], 1)
    ptj = np.concatenate([rng.normal(size=(n, 2)) * rng.choice([0.1, 1.0, 10.0], (n, 1)), np.ones((n, 1))], 1)
    r, Js = tcv.eval_proj(np.concatenate([pti, ptj], 1), np.concatenate([pi, pj, ex, lam[:, None]], 1), synth.PROJ_SQRT_INFO)
    w = dict(r=0.0, J=0.0); nb = 0
    for k in range(n):
        rr, JJ = npo.proj_evaluate(pi[k], pj[k], ex[k], float(lam[k]), pti[k], ptj[k], synth.PROJ_SQRT_INFO)
        e = [rel(r[k], rr)] + [rel(Js[b][k], JJ[b]) for b in range(4)]
        if not (np.all(np.isfinite(rr)) and all(np.all(np.isfinite(J)) for J in JJ)):
            continue      # (a point exactly on the horizon of camera j: both sides overflow)
        ok = e[0] < 1e-10 and max(e[1:]) < 1e-10
        w["r"] = max(w["r"], e[0]); w["J"] = max(w["J"], max(e[1:])); nb += 0 if ok else 1
        if not ok and nb <= 5:
            print("  P1 case", k, [f"{x:.1e}" for x in e], "lam", lam[k])
    print(f"ProjectionFactor: {n} cases, worst r {w['r']:.1e} J {w['J']:.1e}, beyond 1e-10: {nb}"); bad += nb
    # ---- L1
    ln = synth.window_at(synth.make_windows(1000, 1), 0)["line"]
    po = np.array([rand_pose(rng, float(rng.choice([0.1, 2.0, 30.0])), unit=rng.random() < 0.5) for _ in range(n)])
    ps = rng.normal(size=(n, 3)) * rng.choice([1.0, 10.0, 100.0], (n, 1)); pe = ps + rng.normal(size=(n, 3)) * rng.choice([0.1, 1.0, 10.0], (n, 1))
    abc = rng.normal(size=(n, 3)); abc[:, :2] /= np.linalg.norm(abc[:, :2], axis=1, keepdims=True); abc[:, 2] *= 200
    r, J = tcv.eval_line(np.concatenate([ps, pe, abc], 1), ln["K"], ln["Ric"], ln["Tic"], po)
    w = dict(r=0.0, J=0.0); nb = 0
    for k in range(n):
        with np.errstate(all="ignore"):
            rr, JJ = npo.line_evaluate(po[k], ps[k], pe[k], abc[k], ln["K"], ln["Ric"], ln["Tic"])
        if not (np.all(np.isfinite(rr)) and np.all(np.isfinite(JJ[0]))):
            continue
        e = [rel(r[k], rr), rel(J[k], JJ[0])]
        ok = e[0] < 1e-10 and e[1] < 1e-10
        w["r"] = max(w["r"], e[0]); w["J"] = max(w["J"], e[1]); nb += 0 if ok else 1
        if not ok and nb <= 5:
            print("  L1 case", k, [f"{x:.1e}" for x in e])
    print(f"LineProjectionFactor: {n} cases, worst r {w['r']:.1e} J {w['J']:.1e}, beyond 1e-10: {nb}"); bad += nb
    # ---- I1 (the pre-integration itself comes from the oracle: random streams of 2 .. 2000 samples)
    m = max(1, n // 10)
    keys = ["delta_p", "delta_q", "delta_v", "lin_ba", "lin_bg", "sum_dt", "jacobian", "covariance"]
    imu = {k: [] for k in keys}; params = []; G = np.array([0, 0, 9.81])
    refs = []
    for k in range(m):
        S = int(rng.choice([2, 20, 200, 2000]))
        acc = rng.normal(size=(S + 1, 3)) * float(rng.choice([0.5, 5.0])) + G; gyr = rng.normal(size=(S + 1, 3)) * float(rng.choice([0.1, 2.0]))
        ba = rng.normal(size=3) * 0.05; bg = rng.normal(size=3) * 0.005
        pre = npo.preintegrate(acc, gyr, 0.005, ba, bg, *NOISE)
        pre = dict(pre, lin_ba=ba, lin_bg=bg)
        a = rand_pose(rng, 3.0, unit=rng.random() < 0.6); b = rand_pose(rng, 3.0, unit=rng.random() < 0.6)
        sa = np.concatenate([rng.normal(size=3) * 2, ba + rng.normal(size=3) * float(rng.choice([0.0, 0.01, 0.3])), bg + rng.normal(size=3) * float(rng.choice([0.0, 0.001, 0.05]))])
        sb = np.concatenate([rng.normal(size=3) * 2, rng.normal(size=3) * 0.1, rng.normal(size=3) * 0.01])
        Sq = npo.imu_sqrt_info(pre["covariance"])
        rr, JJ = npo.imu_evaluate(a, sa, b, sb, pre, G, sqrt_info=Sq)
        for kk in keys:
            imu[kk].append(pre[kk])
        params.append(np.concatenate([a, sa, b, sb])); refs.append((rr, JJ, Sq))
    imu = {k: np.array(v) for k, v in imu.items()}; imu["frame_i"] = np.zeros(m, int)
    r, Js, _ = tcv.eval_imu(imu, np.array(params), G, sqrt_info=np.array([x[2] for x in refs]))
    w = dict(r=0.0, J=0.0); nb = 0
    for k in range(m):
        rr, JJ, _ = refs[k]
        e = [rel(r[k], rr)] + [rel(Js[b][k], JJ[b]) for b in range(4)]
        ok = e[0] < 1e-9 and max(e[1:]) < 1e-9
        w["r"] = max(w["r"], e[0]); w["J"] = max(w["J"], max(e[1:])); nb += 0 if ok else 1
        if not ok and nb <= 5:
            print("  I1 case", k, [f"{x:.1e}" for x in e], "sum_dt", imu["sum_dt"][k])
    print(f"IMUFactor: {m} cases, worst r {w['r']:.1e} J {w['J']:.1e}, beyond 1e-9: {nb}"); bad += nb
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
