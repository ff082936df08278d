"""Test helper: a relocalisation set-up on a synthetic window, the way `Estimator::setReloFrame` + `OptimizationWithLine` build it
(estimator.cpp:2261-2280, :1854-1886): the loop-closure frame matched window frame f; `relo_Pose` starts as a copy of `para_Pose[f]`
(:2276; here slightly perturbed, the old pose of the loop frame is not the current estimate); every landmark that starts at or before f
and was matched gets one more ProjectionFactor(pts_i, pts_j) on (para_Pose[start], relo_Pose, para_Ex_Pose[0], para_Feature[idx]) with
pts_i its first observation and pts_j the matched point of the loop-closure frame (here: what frame f itself saw, plus pixel noise)."""
import numpy as np


def add_relocalisation(win, f=4, seed=0, sigma_px=1.0 / 460.0):
    rng = np.random.default_rng(seed)
    pr = win["proj"]
    fi, fj, lm = np.asarray(pr["frame_i"]), np.asarray(pr["frame_j"]), np.asarray(pr["landmark"])
    start = {}
    for k in range(len(fi)):
        start.setdefault(int(lm[k]), (int(fi[k]), np.asarray(pr["pts_i"][k])))
    frame_i, landmark, pts_i, pts_j = [], [], [], []
    for k in range(len(fi)):
        l = int(lm[k])
        if int(fj[k]) == f and start[l][0] <= f:          # the landmark is seen in frame f: the loop-closure frame "matches" it
            frame_i.append(start[l][0]); landmark.append(l); pts_i.append(start[l][1])
            pj = np.asarray(pr["pts_j"][k]).copy(); pj[:2] += rng.normal(size=2) * sigma_px
            pts_j.append(pj)
    pose = np.asarray(win["pose"][f], dtype=float).copy()
    pose[:3] += rng.normal(size=3) * 0.03
    dq = np.concatenate([rng.normal(size=3) * 0.004, [1.0]])
    q = pose[3:]
    x, y, z, w = q; a, b, c, d = dq                      # q (x) dq, xyzw
    pose[3:] = np.array([w * a + x * d + y * c - z * b, w * b - x * c + y * d + z * a, w * c + x * b - y * a + z * d, w * d - x * a - y * b - z * c])
    pose[3:] /= np.linalg.norm(pose[3:])
    out = dict(win)
    out["relo"] = dict(pose=pose, frame_i=np.array(frame_i, int), landmark=np.array(landmark, int), pts_i=np.array(pts_i).reshape(-1, 3), pts_j=np.array(pts_j).reshape(-1, 3))
    return out
