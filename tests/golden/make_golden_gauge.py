"""Generates tests/golden/gauge.npz: known-answer vectors for the post-solve gauge fix
(Estimator::double2vector, reference vins_estimator/src/estimator.cpp:1537-1581) from the NumPy oracle.

    python tests/golden/make_golden_gauge.py

Cases: random drifted windows, zero drift, yaw drift across +-180 deg, origin / solved pose within 1 deg of the Euler
singularity (the `rot_diff = Rs[0] * R00^T` branch, :1555-1563), rotations with negative trace (every branch of the
matrix -> quaternion conversion of vector2double, :1499)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import np_oracle as O  # noqa: E402


def rand_q(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def make_case(rng, kind, n=11):
    pose = np.zeros((n, 7)); sb = rng.normal(size=(n, 9))
    for i in range(n):
        pose[i, :3] = rng.normal(size=3) * 3.0
        pose[i, 3:] = rand_q(rng) * (1.0 + 1e-9 * rng.normal())      # solver output is unit only to rounding
    q0 = rand_q(rng)
    if kind == "identity":
        q0 = pose[0, 3:] / np.linalg.norm(pose[0, 3:])
    elif kind == "wrap":
        pose[0, 3:] = O.R2q(O.ypr2R([179.0, 5.0, -3.0])); q0 = O.R2q(O.ypr2R([-178.5, 4.0, -2.0]))
    elif kind == "singular_origin":
        q0 = O.R2q(O.ypr2R([30.0, 89.6, 10.0]))
    elif kind == "singular_solved":
        pose[0, 3:] = O.R2q(O.ypr2R([-60.0, -90.4, 20.0]))
    elif kind == "negtrace":
        for i in range(n):
            pose[i, 3:] = O.R2q(O.ypr2R([170.0 - 30.0 * i, 100.0 + 7.0 * i, 160.0 + 11.0 * i]))
    R0 = O.q2R(q0)
    P0 = rng.normal(size=3)
    return R0, P0, pose, sb


def main():
    rng = np.random.default_rng(20251001)
    kinds = ["random"] * 10 + ["identity", "wrap", "singular_origin", "singular_solved", "negtrace", "negtrace"]
    out = {k: [] for k in ("R0", "P0", "pose", "sb", "Rs", "Ps", "Vs", "pose_out")}
    for kind in kinds:
        R0, P0, pose, sb = make_case(rng, kind)
        Rs, Ps, Vs, po = O.gauge_fix(R0, P0, pose, sb)
        for k, v in zip(out, (R0, P0, pose, sb, Rs, Ps, Vs, po)):
            out[k].append(v)
    np.savez_compressed(os.path.join(HERE, "gauge.npz"), kinds=np.array(kinds), **{k: np.array(v) for k, v in out.items()})
    print("wrote gauge.npz:", len(kinds), "cases")


if __name__ == "__main__":
    main()
