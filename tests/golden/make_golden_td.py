"""Generates tests/golden/proj_td.npz: known-answer vectors for ProjectionTdFactor::Evaluate
(reference vins_estimator/src/factor/projection_td_factor.cpp:34-140) from the NumPy oracle.

    python tests/golden/make_golden_td.py

32 random cases (two poses, extrinsic, inverse depth, td; feature velocities, per-frame td, image rows) with a rolling-shutter
read-out time TR = 0.03 s over ROW = 480 lines, plus global-shutter cases (TR = 0) and zero feature velocity (the factor
degenerates to ProjectionFactor with a zero td column)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import np_oracle as O  # noqa: E402


def main():
    rng = np.random.default_rng(7101)
    n = 40
    pts = np.zeros((n, 6)); aux = np.zeros((n, 8)); params = np.zeros((n, 23)); TRs = np.zeros(n)
    res = np.zeros((n, 2)); jacs = [np.zeros((n, 2, 7)), np.zeros((n, 2, 7)), np.zeros((n, 2, 7)), np.zeros((n, 2, 1)), np.zeros((n, 2, 1))]
    ROW = 480.0
    for k in range(n):
        qi = rng.normal(size=4); qi /= np.linalg.norm(qi)
        dq = np.concatenate([rng.normal(size=3) * 0.05, [1.0]]); qj = O.qmul(qi, dq); qj /= np.linalg.norm(qj)
        qe = np.concatenate([rng.normal(size=3) * 0.02, [1.0]]); qe /= np.linalg.norm(qe)
        Pi = rng.normal(size=3); Pj = Pi + rng.normal(size=3) * 0.2; tic = rng.normal(size=3) * 0.05
        depth = rng.uniform(2.0, 8.0)
        pi = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.4, 0.4), 1.0])
        pw = O.qrot(qi, O.qrot(qe, pi * depth) + tic) + Pi
        pcj = O.qrot(O.qinv(qe), O.qrot(O.qinv(qj), pw - Pj) - tic)
        pj = np.array([pcj[0] / pcj[2], pcj[1] / pcj[2], 1.0]) + np.array([rng.normal() * 0.002, rng.normal() * 0.002, 0.0])
        pts[k] = np.concatenate([pi, pj])
        aux[k] = [rng.normal() * 0.3, rng.normal() * 0.3, rng.normal() * 0.3, rng.normal() * 0.3, rng.normal() * 0.005, rng.normal() * 0.005,
                  rng.uniform(0, ROW), rng.uniform(0, ROW)]
        TRs[k] = 0.03 if k < 32 else 0.0
        if k >= 36:
            aux[k, :4] = 0.0
        params[k] = np.concatenate([Pi, qi, Pj, qj, tic, qe, [1.0 / depth * (1 + rng.uniform(-0.1, 0.1))], [rng.normal() * 0.01]])
        r, Js = O.proj_td_evaluate(params[k, 0:7], params[k, 7:14], params[k, 14:21], params[k, 21], params[k, 22], pts[k, :3], pts[k, 3:],
                                   aux[k, 0:2], aux[k, 2:4], aux[k, 4], aux[k, 5], aux[k, 6], aux[k, 7], 460.0 / 1.5, TRs[k], ROW)
        res[k] = r
        for b in range(5):
            jacs[b][k] = Js[b]
    np.savez_compressed(os.path.join(HERE, "proj_td.npz"), pts=pts, aux=aux, params=params, TR=TRs, ROW=ROW, sqrt_info=460.0 / 1.5, res=res,
                        J_pose_i=jacs[0], J_pose_j=jacs[1], J_ex=jacs[2], J_lam=jacs[3], J_td=jacs[4])
    print("wrote proj_td.npz:", n, "cases")


if __name__ == "__main__":
    main()
