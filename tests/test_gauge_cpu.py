"""G3: the post-solve gauge fix Estimator::double2vector (estimator.cpp:1537-1581).
CPU side: the C oracle against the committed known-answer vectors, plus the properties the reference relies on."""
import numpy as np

import np_oracle as O
import orc
from util import load, rel


def test_c_oracle_gauge_fix_matches_golden():
    z = load("gauge.npz")
    for k in range(len(z["kinds"])):
        Rs, Ps, Vs, po = orc.gauge_fix(z["R0"][k], z["P0"][k], z["pose"][k], z["sb"][k])
        assert rel(Rs, z["Rs"][k]) < 1e-13 and rel(Ps, z["Ps"][k]) < 1e-13 and rel(Vs, z["Vs"][k]) < 1e-13, z["kinds"][k]
        assert rel(po, z["pose_out"][k]) < 1e-12, z["kinds"][k]


def test_gauge_fix_properties():
    """frame 0 is re-anchored at the original position with the original yaw; relative geometry is untouched."""
    z = load("gauge.npz")
    for k in range(len(z["kinds"])):
        Rs, Ps, Vs, po = z["Rs"][k], z["Ps"][k], z["Vs"][k], z["pose_out"][k]
        pose = z["pose"][k]
        assert np.allclose(Ps[0], z["P0"][k], atol=1e-14)
        if not str(z["kinds"][k]).startswith("singular"):
            y0 = O.R2ypr(z["R0"][k])[0]; y1 = O.R2ypr(Rs[0])[0]
            assert abs(((y1 - y0) + 180.0) % 360.0 - 180.0) < 1e-6        # R00 is formed without normalising the (1 +- 1e-9)-norm quaternion, like :1548
        else:
            assert np.allclose(Rs[0], z["R0"][k], atol=1e-9)          # full rotation reset near the Euler singularity (:1555-1563)
        for i in range(1, pose.shape[0]):
            Ri = O.q2R(O.qnormalized(pose[i, 3:])); R0s = O.q2R(O.qnormalized(pose[0, 3:]))
            assert np.allclose(Rs[0].T @ Rs[i], R0s.T @ Ri, atol=1e-8)
            assert abs(np.linalg.norm(Ps[i] - Ps[0]) - np.linalg.norm(pose[i, :3] - pose[0, :3])) < 1e-7   # rot_diff inherits the 1e-9 non-unit norm of q0 in the singular branch
            assert abs(np.linalg.norm(Vs[i]) - np.linalg.norm(z["sb"][k][i, :3])) < 1e-7
        for i in range(pose.shape[0]):                                # the quaternion written by vector2double represents Rs[i]
            assert np.allclose(O.q2R(po[i, 3:]), Rs[i], atol=1e-8) and abs(np.linalg.norm(po[i, 3:]) - 1.0) < 1e-8


def test_gauge_fix_is_identity_without_drift():
    z = load("gauge.npz")
    k = list(z["kinds"]).index("identity")
    pose = z["pose"][k]
    Rs, Ps, Vs, po = orc.gauge_fix(z["R0"][k], pose[0, :3], pose, z["sb"][k])
    assert np.allclose(Ps, pose[:, :3], atol=1e-12) and np.allclose(Vs, z["sb"][k][:, :3], atol=1e-12)
