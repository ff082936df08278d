"""Trajectory files and the absolute trajectory error.

* `write_vins_result` / `read_vins_result`: the CSV the reference appends per frame in `pubOdometry`
  (vins_estimator/src/utility/visualization.cpp:210-226): `t[ns], px, py, pz, qw, qx, qy, qz, vx, vy, vz,` with
  `fixed` notation, precision 0 for the stamp and 5 for the rest, and the trailing comma.
* `read_euroc_groundtruth`: the ground-truth rows the reference's benchmark_publisher parses
  (benchmark_publisher/src/benchmark_publisher_node.cpp:42-62): 17 comma-separated columns
  `t[ns], p(3), q(wxyz), v(3), bw(3), ba(3)`, one header line.
* `ate_rmse`: RMSE of the positions after a least-squares SE(3) (or Sim(3)) alignment (Umeyama 1991), the ATE the
  paper's Table II reports.

Host-side evaluation only (NumPy); nothing here is on the solver's hot path.
"""
from __future__ import annotations

import numpy as np


def write_vins_result(path, stamps_s, Ps, quats_xyzw, Vs, append=False):
    """one row per frame in the reference's format; quaternions are given x y z w and written w x y z."""
    with open(path, "a" if append else "w") as f:
        for t, p, q, v in zip(stamps_s, Ps, quats_xyzw, Vs):
            f.write("%.0f," % (t * 1e9))
            f.write(",".join("%.5f" % x for x in (p[0], p[1], p[2], q[3], q[0], q[1], q[2], v[0], v[1], v[2])) + ",\n")


def read_vins_result(path):
    rows = []
    with open(path) as f:
        for line in f:
            parts = [x for x in line.strip().split(",") if x != ""]
            if len(parts) >= 11:
                rows.append([float(x) for x in parts[:11]])
    a = np.array(rows).reshape(-1, 11)
    return dict(t=a[:, 0] * 1e-9, p=a[:, 1:4], q_xyzw=a[:, [5, 6, 7, 4]], v=a[:, 8:11])


def read_euroc_groundtruth(path):
    a = np.loadtxt(path, delimiter=",", skiprows=1, usecols=range(17), ndmin=2)
    return dict(t=a[:, 0] * 1e-9, p=a[:, 1:4], q_xyzw=a[:, [5, 6, 7, 4]], v=a[:, 8:11], bw=a[:, 11:14], ba=a[:, 14:17])


def associate(t_est, t_ref, max_dt=0.01):
    """nearest-stamp association; returns index arrays (est, ref)."""
    t_est = np.asarray(t_est, dtype=float); t_ref = np.asarray(t_ref, dtype=float)
    j = np.clip(np.searchsorted(t_ref, t_est), 1, len(t_ref) - 1)
    j = np.where(np.abs(t_ref[j - 1] - t_est) <= np.abs(t_ref[j] - t_est), j - 1, j)
    ok = np.abs(t_ref[j] - t_est) <= max_dt
    return np.nonzero(ok)[0], j[ok]


def umeyama(src, dst, with_scale=False):
    """least-squares similarity dst ~ s R src + t (Umeyama 1991); returns (s, R, t)."""
    src = np.asarray(src, dtype=float); dst = np.asarray(dst, dtype=float)
    mu_s, mu_d = src.mean(0), dst.mean(0)
    xs, xd = src - mu_s, dst - mu_d
    cov = xd.T @ xs / src.shape[0]
    U, D, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1.0
    R = U @ S @ Vt
    s = float(np.trace(np.diag(D) @ S) / (xs ** 2).sum() * src.shape[0]) if with_scale else 1.0
    t = mu_d - s * R @ mu_s
    return s, R, t


def ate_rmse(p_est, p_ref, align=True, with_scale=False):
    """absolute trajectory error: RMSE of |s R p_est + t - p_ref| (metres).  align=False compares the raw positions."""
    p_est = np.asarray(p_est, dtype=float); p_ref = np.asarray(p_ref, dtype=float)
    if align and len(p_est) >= 3:
        s, R, t = umeyama(p_est, p_ref, with_scale)
        p_est = s * p_est @ R.T + t
    d = np.linalg.norm(p_est - p_ref, axis=1)
    return float(np.sqrt((d ** 2).mean()))
