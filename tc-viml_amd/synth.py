"""Deterministic synthetic sliding-window generator.

Produces batches of independent windows shaped like the ones
`Estimator::OptimizationWithLine` (reference vins_estimator/src/estimator.cpp:1677-1900)
hands to Ceres: 11 keyframes @10 Hz, 10 IMU pre-integrations (200 Hz mid-point
integration, integration_base.h:54-158), 200 point-reprojection blocks over 50
landmarks (consecutive tracks starting at the anchor frame, estimator.cpp:1737-1771),
40 2D-3D line blocks (estimator.cpp:1786-1846) and, optionally, a marginalisation
prior.  Everything is FP64 and batched over the leading axis `B`; the RNG of
window `w` is `PCG64(0xC0FFEE + w)` so any window can be regenerated in isolation.

This file is workload generation (host side, NumPy).  It deliberately carries its
own mid-point pre-integration so it never touches `oracle/`.
"""
from __future__ import annotations

import os
import numpy as np

# ----------------------------------------------------------------------------------------
# constants mirrored from the reference configuration
# (benchmark_publisher/config/V1_01_easy/sensor.yaml; parameters.h:19-22)
# ----------------------------------------------------------------------------------------
WINDOW_SIZE = 10
N_FRAMES = WINDOW_SIZE + 1
FOCAL_LENGTH = 460.0
PROJ_SQRT_INFO = FOCAL_LENGTH / 1.5          # estimator.cpp:48,85
ACC_N, GYR_N, ACC_W, GYR_W = 0.08, 0.004, 0.00004, 2.0e-6   # sensor.yaml:90-93
G_NORM = 9.81007                                             # sensor.yaml:94
FX, FY, CX, CY = 461.6, 460.3, 363.0, 248.1                  # sensor.yaml:16-19
IMG_W, IMG_H = 752.0, 480.0
RIC = np.array([[0.0148655429818, -0.999880929698, 0.00414029679422],
                [0.999557249008, 0.0149672133247, 0.025715529948],
                [-0.0257744366974, 0.00375618835797, 0.999660727178]])   # sensor.yaml:61-67
TIC = np.array([-0.0216401454975, -0.064676986768, 0.00981073058949])    # sensor.yaml:69-73
RBW = np.array([[0.958882, 0.283788, -0.00258614],
                [-0.283713, 0.958774, 0.016038],
                [0.00703105, -0.0146448, 0.999868]])                     # sensor.yaml:40-49
TBW = np.array([-1.4494, -1.83337, -0.899281])                           # sensor.yaml:52-57
K_MAT = np.array([[FX, 0.0, CX], [0.0, FY, CY], [0.0, 0.0, 1.0]])

IMU_RATE_SUB = 20            # IMU samples per keyframe interval (200 Hz / 10 Hz)
DT_IMU = 0.005
DT_KF = 0.1

N_LANDMARKS = 50
N_LINES = 40

# later-observation counts per landmark (sum = 200).  Landmark k is anchored in frame
# k % 7; tracks are consecutive frames from the anchor as in estimator.cpp:1745-1770.
_TRACKS = {0: [10, 2, 2, 3, 4, 3, 4, 4], 1: [9, 2, 2, 3, 4, 4, 4], 2: [8, 2, 3, 3, 4, 4, 4],
           3: [7, 2, 3, 4, 4, 4, 4], 4: [6, 3, 3, 4, 4, 4, 4], 5: [5, 3, 4, 4, 4, 4, 4],
           6: [4, 4, 4, 4, 4, 4, 4]}


def track_table(n_landmarks: int = N_LANDMARKS):
    """(anchor frame, number of later observations) for every landmark."""
    anchors, counts = [], []
    seen = {i: 0 for i in range(7)}
    for k in range(n_landmarks):
        a = k % 7
        c = _TRACKS[a][seen[a] % len(_TRACKS[a])]
        seen[a] += 1
        anchors.append(a)
        counts.append(min(c, WINDOW_SIZE - a))
    return np.array(anchors), np.array(counts)


# ----------------------------------------------------------------------------------------
# small batched quaternion / rotation helpers (quaternions stored x,y,z,w like
# Eigen::Map<Quaterniond>, pose_local_parameterization.cpp:6)
# ----------------------------------------------------------------------------------------
def qmul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def qrot(q, v):
    u = q[..., :3]
    w = q[..., 3:4]
    uv = np.cross(u, v)
    uv = uv + uv
    return v + w * uv + np.cross(u, uv)


def qnormalize(q):
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def q2R(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - (tyy + tzz); R[..., 0, 1] = txy - twz; R[..., 0, 2] = txz + twy
    R[..., 1, 0] = txy + twz; R[..., 1, 1] = 1 - (txx + tzz); R[..., 1, 2] = tyz - twx
    R[..., 2, 0] = txz - twy; R[..., 2, 1] = tyz + twx; R[..., 2, 2] = 1 - (txx + tyy)
    return R


def R2q(R):
    """Rotation matrix -> quaternion (x,y,z,w), w >= 0 branch selection as usual."""
    R = np.asarray(R)
    out = np.empty(R.shape[:-2] + (4,))
    flat_R = R.reshape(-1, 3, 3)
    flat_o = out.reshape(-1, 4)
    for n in range(flat_R.shape[0]):
        m = flat_R[n]
        t = m[0, 0] + m[1, 1] + m[2, 2]
        if t > 0:
            s = np.sqrt(t + 1.0) * 2
            q = [(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s]
        else:
            i = int(np.argmax([m[0, 0], m[1, 1], m[2, 2]]))
            j, k = (i + 1) % 3, (i + 2) % 3
            s = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0) * 2
            q = [0.0, 0.0, 0.0, 0.0]
            q[i] = 0.25 * s
            q[j] = (m[j, i] + m[i, j]) / s
            q[k] = (m[k, i] + m[i, k]) / s
            q[3] = (m[k, j] - m[j, k]) / s
        flat_o[n] = q
    return out


def skew(v):
    S = np.zeros(v.shape[:-1] + (3, 3))
    S[..., 0, 1] = -v[..., 2]; S[..., 0, 2] = v[..., 1]
    S[..., 1, 0] = v[..., 2]; S[..., 1, 2] = -v[..., 0]
    S[..., 2, 0] = -v[..., 1]; S[..., 2, 1] = v[..., 0]
    return S


# ----------------------------------------------------------------------------------------
# analytic trajectory
# ----------------------------------------------------------------------------------------
_R_FIX = np.array([[0.0, 0.0, 1.0], [0.0, -1.0, 0.0], [1.0, 0.0, 0.0]])   # body z forward, x up


def _Rz(a):
    c, s = np.cos(a), np.sin(a)
    R = np.zeros(a.shape + (3, 3)); R[..., 0, 0] = c; R[..., 0, 1] = -s; R[..., 1, 0] = s; R[..., 1, 1] = c; R[..., 2, 2] = 1
    return R


def _Ry(a):
    c, s = np.cos(a), np.sin(a)
    R = np.zeros(a.shape + (3, 3)); R[..., 0, 0] = c; R[..., 0, 2] = s; R[..., 1, 1] = 1; R[..., 2, 0] = -s; R[..., 2, 2] = c
    return R


def _Rx(a):
    c, s = np.cos(a), np.sin(a)
    R = np.zeros(a.shape + (3, 3)); R[..., 0, 0] = 1; R[..., 1, 1] = c; R[..., 1, 2] = -s; R[..., 2, 1] = s; R[..., 2, 2] = c
    return R


def traj_R(t):
    yaw = 0.5 * t + np.pi / 2
    roll = np.deg2rad(5.0) * np.sin(0.7 * t)
    pitch = np.deg2rad(5.0) * np.sin(0.9 * t + 0.3)
    return _Rz(yaw) @ _Ry(pitch) @ _Rx(roll) @ _R_FIX


def traj_p(t):
    return np.stack([2 * np.cos(0.5 * t), 2 * np.sin(0.5 * t), 0.3 * np.sin(t)], -1)


def traj_v(t):
    return np.stack([-np.sin(0.5 * t), np.cos(0.5 * t), 0.3 * np.cos(t)], -1)


def traj_a(t):
    return np.stack([-0.5 * np.cos(0.5 * t), -0.5 * np.sin(0.5 * t), -0.3 * np.sin(t)], -1)


def traj_w_body(t, h=1e-6):
    R = traj_R(t)
    dR = (traj_R(t + h) - traj_R(t - h)) / (2 * h)
    W = np.swapaxes(R, -1, -2) @ dR
    return np.stack([W[..., 2, 1], W[..., 0, 2], W[..., 1, 0]], -1)


# ----------------------------------------------------------------------------------------
# mid-point IMU pre-integration, batched (integration_base.h:13-158)
# ----------------------------------------------------------------------------------------
def preintegrate(acc, gyr, dt, ba, bg):
    """acc, gyr: (..., S+1, 3) with sample 0 taken at the start keyframe; dt scalar.
    Returns dict with delta_p, delta_q(xyzw), delta_v, jacobian(15x15), covariance(15x15), sum_dt."""
    lead = acc.shape[:-2]
    S = acc.shape[-2] - 1
    noise = np.zeros((18, 18))
    noise[0:3, 0:3] = ACC_N ** 2 * np.eye(3); noise[3:6, 3:6] = GYR_N ** 2 * np.eye(3)
    noise[6:9, 6:9] = ACC_N ** 2 * np.eye(3); noise[9:12, 9:12] = GYR_N ** 2 * np.eye(3)
    noise[12:15, 12:15] = ACC_W ** 2 * np.eye(3); noise[15:18, 15:18] = GYR_W ** 2 * np.eye(3)
    dp = np.zeros(lead + (3,)); dv = np.zeros(lead + (3,))
    dq = np.zeros(lead + (4,)); dq[..., 3] = 1.0
    jac = np.broadcast_to(np.eye(15), lead + (15, 15)).copy()
    cov = np.zeros(lead + (15, 15))
    I3 = np.eye(3)
    sum_dt = 0.0
    for k in range(S):
        a0, g0, a1, g1 = acc[..., k, :], gyr[..., k, :], acc[..., k + 1, :], gyr[..., k + 1, :]
        un_acc_0 = qrot(dq, a0 - ba)
        un_gyr = 0.5 * (g0 + g1) - bg
        dq_step = np.concatenate([un_gyr * dt / 2, np.ones(lead + (1,))], -1)
        rq = qmul(dq, dq_step)
        un_acc_1 = qrot(rq, a1 - ba)
        un_acc = 0.5 * (un_acc_0 + un_acc_1)
        rp = dp + dv * dt + 0.5 * un_acc * dt * dt
        rv = dv + un_acc * dt
        R_w_x = skew(un_gyr); R_a0 = skew(a0 - ba); R_a1 = skew(a1 - ba)
        R0 = q2R(dq); R1 = q2R(rq)
        F = np.zeros(lead + (15, 15))
        F[..., 0:3, 0:3] = I3
        F[..., 0:3, 3:6] = -0.25 * R0 @ R_a0 * dt * dt + -0.25 * R1 @ R_a1 @ (I3 - R_w_x * dt) * dt * dt
        F[..., 0:3, 6:9] = I3 * dt
        F[..., 0:3, 9:12] = -0.25 * (R0 + R1) * dt * dt
        F[..., 0:3, 12:15] = -0.25 * R1 @ R_a1 * dt * dt * -dt
        F[..., 3:6, 3:6] = I3 - R_w_x * dt
        F[..., 3:6, 12:15] = -1.0 * I3 * dt
        F[..., 6:9, 3:6] = -0.5 * R0 @ R_a0 * dt + -0.5 * R1 @ R_a1 @ (I3 - R_w_x * dt) * dt
        F[..., 6:9, 6:9] = I3
        F[..., 6:9, 9:12] = -0.5 * (R0 + R1) * dt
        F[..., 6:9, 12:15] = -0.5 * R1 @ R_a1 * dt * -dt
        F[..., 9:12, 9:12] = I3
        F[..., 12:15, 12:15] = I3
        V = np.zeros(lead + (15, 18))
        V[..., 0:3, 0:3] = 0.25 * R0 * dt * dt
        V[..., 0:3, 3:6] = 0.25 * -R1 @ R_a1 * dt * dt * 0.5 * dt
        V[..., 0:3, 6:9] = 0.25 * R1 * dt * dt
        V[..., 0:3, 9:12] = V[..., 0:3, 3:6]
        V[..., 3:6, 3:6] = 0.5 * I3 * dt
        V[..., 3:6, 9:12] = 0.5 * I3 * dt
        V[..., 6:9, 0:3] = 0.5 * R0 * dt
        V[..., 6:9, 3:6] = 0.5 * -R1 @ R_a1 * dt * 0.5 * dt
        V[..., 6:9, 6:9] = 0.5 * R1 * dt
        V[..., 6:9, 9:12] = V[..., 6:9, 3:6]
        V[..., 9:12, 12:15] = I3 * dt
        V[..., 12:15, 15:18] = I3 * dt
        jac = F @ jac
        cov = F @ cov @ np.swapaxes(F, -1, -2) + V @ noise @ np.swapaxes(V, -1, -2)
        dp, dv = rp, rv
        dq = qnormalize(rq)
        sum_dt += dt
    return dict(delta_p=dp, delta_q=dq, delta_v=dv, jacobian=jac, covariance=cov,
                sum_dt=np.full(lead, sum_dt))


# ----------------------------------------------------------------------------------------
# 3D line pool (data fixture: subset of the reference's V1_01 prior line map)
# ----------------------------------------------------------------------------------------
_LINE_POOL = None


def line_pool():
    global _LINE_POOL
    if _LINE_POOL is None:
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.path.join(here, "data", "lines3d_v101_subset.txt")
        raw = np.loadtxt(path)
        ps = raw[:, 0:3] @ RBW.T + TBW        # estimator.cpp:1832-1833
        pe = raw[:, 3:6] @ RBW.T + TBW
        _LINE_POOL = (ps, pe)
    return _LINE_POOL


# ----------------------------------------------------------------------------------------
# window generation
# ----------------------------------------------------------------------------------------
def _cam_pose(Rwb, pwb):
    """world<-camera rotation/translation from world<-body."""
    Rwc = Rwb @ RIC
    twc = pwb + (Rwb @ TIC[..., None])[..., 0]
    return Rwc, twc


def _project(Rwc, twc, Pw):
    """Pw (...,3) into camera with pose (...,3,3),(...,3) -> camera coords."""
    d = Pw - twc
    return (np.swapaxes(Rwc, -1, -2) @ d[..., None])[..., 0]


def _visible_norm(pc, margin=0.0):
    z = pc[..., 2]
    x = pc[..., 0] / np.where(z > 1e-9, z, 1.0)
    y = pc[..., 1] / np.where(z > 1e-9, z, 1.0)
    u = FX * x + CX
    v = FY * y + CY
    return (z > 0.5) & (u > margin) & (u < IMG_W - margin) & (v > margin) & (v < IMG_H - margin)


def make_windows(first_id: int, count: int, *, n_landmarks: int = N_LANDMARKS, n_lines: int = N_LINES,
                 frame_shift: int = 0, with_lines: bool = True):
    """Generate `count` independent windows with ids first_id ... first_id+count-1.

    `frame_shift = -1` generates the "pre-window" (keyframes -1..9 of the same trajectory
    segment) whose marginalisation yields the prior of the main window (frame_shift = 0).
    Returns a dict of batched arrays (leading axis B)."""
    B = count
    ids = np.arange(first_id, first_id + count)
    F = N_FRAMES
    t0 = 1.0 + 0.731 * (ids % 4096) + frame_shift * DT_KF
    tk = t0[:, None] + DT_KF * np.arange(F)[None, :]                    # (B,F)
    Rwb = traj_R(tk); pwb = traj_p(tk); vwb = traj_v(tk)
    G = np.array([0.0, 0.0, G_NORM])

    # per-window random draws (one PCG64 stream per window id, independent of batch shape
    # and of frame_shift so that pre-window and main window share biases)
    ba_true = np.empty((B, 3)); bg_true = np.empty((B, 3))
    imu_noise = np.empty((B, F + 1, IMU_RATE_SUB, 6))       # indexed by absolute interval (kf -1..)
    lm_rand = np.empty((B, 2, n_landmarks, 32, 3))          # [shift slot]
    lm_obs_noise = np.empty((B, 2, n_landmarks, F, 2))
    lm_depth_pert = np.empty((B, 2, n_landmarks))
    ln_noise = np.empty((B, 2, max(n_lines, 1), 4))
    ln_rand = np.empty((B, 2, max(n_lines, 1), 8))
    st_pert = np.empty((B, F + 1, 9))
    for b, wid in enumerate(ids):
        rng = np.random.Generator(np.random.PCG64(0xC0FFEE + int(wid)))
        ba_true[b] = rng.uniform(-0.05, 0.05, 3)
        bg_true[b] = rng.uniform(-0.01, 0.01, 3)
        imu_noise[b] = rng.standard_normal((F + 1, IMU_RATE_SUB, 6))
        lm_rand[b] = rng.uniform(0.0, 1.0, (2, n_landmarks, 32, 3))
        lm_obs_noise[b] = rng.standard_normal((2, n_landmarks, F, 2))
        lm_depth_pert[b] = rng.uniform(-0.1, 0.1, (2, n_landmarks))
        ln_noise[b] = rng.standard_normal((2, max(n_lines, 1), 4))
        ln_rand[b] = rng.uniform(0.0, 1.0, (2, max(n_lines, 1), 8))
        st_pert[b] = rng.standard_normal((F + 1, 9))
    slot = 0 if frame_shift == 0 else 1
    kf_abs = np.arange(F) + frame_shift + 1           # index into the (F+1)-long absolute tables

    # ---- IMU samples and pre-integration for intervals (k -> k+1), k = 0..F-2
    sub = np.arange(IMU_RATE_SUB + 1) * DT_IMU
    ts = tk[:, :-1, None] + sub[None, None, :]                         # (B,F-1,S+1)
    Rs = traj_R(ts)
    acc = (np.swapaxes(Rs, -1, -2) @ (traj_a(ts) + G)[..., None])[..., 0] + ba_true[:, None, None, :]
    gyr = traj_w_body(ts) + bg_true[:, None, None, :]
    # measurement noise: sample j of interval k is the same physical sample as sample 0 of
    # interval k+1 at j = S, so draw noise per absolute (interval, sub-sample) and share it.
    nz = imu_noise[:, kf_abs[:-1]]                                      # (B,F-1,S,6) for samples 1..S
    nz_prev_last = imu_noise[:, kf_abs[:-1] - 1][:, :, -1, :]           # sample 0 == previous interval's last
    acc[:, :, 1:, :] += ACC_N * nz[..., 0:3]
    gyr[:, :, 1:, :] += GYR_N * nz[..., 3:6]
    acc[:, :, 0, :] += ACC_N * nz_prev_last[..., 0:3]
    gyr[:, :, 0, :] += GYR_N * nz_prev_last[..., 3:6]
    lin_ba = np.zeros((B, F - 1, 3)); lin_bg = np.zeros((B, F - 1, 3))
    pre = preintegrate(acc, gyr, DT_IMU, lin_ba, lin_bg)
    imu = dict(frame_i=np.arange(F - 1), frame_j=np.arange(1, F),
               delta_p=pre["delta_p"], delta_q=pre["delta_q"], delta_v=pre["delta_v"],
               lin_ba=lin_ba, lin_bg=lin_bg, sum_dt=pre["sum_dt"],
               jacobian=pre["jacobian"], covariance=pre["covariance"],
               acc=acc, gyr=gyr)

    # ---- landmarks / point observations
    anchors, counts = track_table(n_landmarks)
    Rwc, twc = _cam_pose(Rwb, pwb)                                      # (B,F,3,3),(B,F,3)
    P_w = np.empty((B, n_landmarks, 3))
    for l in range(n_landmarks):
        a, c = int(anchors[l]), int(counts[l])
        mid = a + c // 2
        cand = lm_rand[:, slot, l]                                      # (B,32,3)
        x = (cand[..., 0] - 0.5) * 0.9
        y = (cand[..., 1] - 0.5) * 0.6
        dist = 3.0 + 5.0 * cand[..., 2]
        ray = np.stack([x, y, np.ones_like(x)], -1)
        pc = ray / np.linalg.norm(ray, axis=-1, keepdims=True) * dist[..., None]
        pw = (Rwc[:, mid][:, None] @ pc[..., None])[..., 0] + twc[:, mid][:, None]   # (B,32,3)
        ok = np.ones(pw.shape[:2], bool)
        for f in range(a, a + c + 1):
            ok &= _visible_norm(_project(Rwc[:, f][:, None], twc[:, f][:, None], pw), margin=8.0)
        first = np.where(ok.any(1), ok.argmax(1), 31)
        P_w[:, l] = pw[np.arange(B), first]
    n_proj = int(counts.sum())
    proj_fi = np.empty(n_proj, int); proj_fj = np.empty(n_proj, int); proj_lm = np.empty(n_proj, int)
    pts_i = np.empty((B, n_proj, 3)); pts_j = np.empty((B, n_proj, 3))
    lam_true = np.empty((B, n_landmarks))
    k = 0
    for l in range(n_landmarks):
        a, c = int(anchors[l]), int(counts[l])
        pc_a = _project(Rwc[:, a], twc[:, a], P_w[:, l])
        lam_true[:, l] = 1.0 / pc_a[:, 2]
        obs_a = np.concatenate([pc_a[:, :2] / pc_a[:, 2:3] + lm_obs_noise[:, slot, l, a] / FOCAL_LENGTH,
                                np.ones((B, 1))], -1)
        for f in range(a + 1, a + c + 1):
            pc_f = _project(Rwc[:, f], twc[:, f], P_w[:, l])
            obs_f = np.concatenate([pc_f[:, :2] / pc_f[:, 2:3] + lm_obs_noise[:, slot, l, f] / FOCAL_LENGTH,
                                    np.ones((B, 1))], -1)
            proj_fi[k], proj_fj[k], proj_lm[k] = a, f, l
            pts_i[:, k], pts_j[:, k] = obs_a, obs_f
            k += 1
    lam0 = lam_true * (1.0 + lm_depth_pert[:, slot])

    # ---- line observations
    q_ex = R2q(RIC)
    Ric_n = q2R(qnormalize(q_ex))           # estimator.cpp:1778-1781 (quaternion normalised, then matrix)
    if with_lines and n_lines > 0:
        pool_s, pool_e = line_pool()
        NP = pool_s.shape[0]
        per_frame = max(1, int(np.ceil(n_lines / WINDOW_SIZE)))
        ln_frame = np.empty(n_lines, int)
        ln_ps = np.empty((B, n_lines, 3)); ln_pe = np.empty((B, n_lines, 3)); ln_abc = np.empty((B, n_lines, 3))
        for i in range(n_lines):
            f = min(i // per_frame, F - 1)
            ln_frame[i] = f
            # candidates from the map pool, rotated start so that different slots pick different lines
            pcs = _project(Rwc[:, f][:, None], twc[:, f][:, None], pool_s[None])     # (B,NP,3)
            pce = _project(Rwc[:, f][:, None], twc[:, f][:, None], pool_e[None])
            ok = _visible_norm(pcs, 4.0) & _visible_norm(pce, 4.0)
            start = (ids * 7 + i * 13) % NP
            order = (start[:, None] + np.arange(NP)[None, :]) % NP                   # (B,NP)
            ok_o = np.take_along_axis(ok, order, 1)
            # the j-th line of a frame takes the j-th visible candidate in rotated order
            jth = i % per_frame
            csum = np.cumsum(ok_o, 1)
            hit = ok_o & (csum == jth + 1)
            has = hit.any(1)
            sel = np.take_along_axis(order, hit.argmax(1)[:, None], 1)[:, 0]
            ps_w = pool_s[sel]; pe_w = pool_e[sel]
            # fall-back: synthesise a segment in front of the camera when the pool has too few visible lines
            r = ln_rand[:, slot, i]
            d1 = 3.0 + 4.0 * r[:, 0]
            p1 = np.stack([(r[:, 1] - 0.5) * 0.8, (r[:, 2] - 0.5) * 0.5, np.ones(B)], -1) * d1[:, None]
            p2 = p1 + np.stack([(r[:, 3] - 0.5) * 1.2, (r[:, 4] - 0.5) * 0.8, (r[:, 5] - 0.5) * 0.6], -1)
            p2[:, 2] = np.maximum(p2[:, 2], 1.0)
            syn_s = (Rwc[:, f] @ p1[..., None])[..., 0] + twc[:, f]
            syn_e = (Rwc[:, f] @ p2[..., None])[..., 0] + twc[:, f]
            ps_w = np.where(has[:, None], ps_w, syn_s)
            pe_w = np.where(has[:, None], pe_w, syn_e)
            ln_ps[:, i], ln_pe[:, i] = ps_w, pe_w
            cs = _project(Rwc[:, f], twc[:, f], ps_w); ce = _project(Rwc[:, f], twc[:, f], pe_w)
            us = FX * cs[:, 0] / cs[:, 2] + CX + ln_noise[:, slot, i, 0]
            vs = FY * cs[:, 1] / cs[:, 2] + CY + ln_noise[:, slot, i, 1]
            ue = FX * ce[:, 0] / ce[:, 2] + CX + ln_noise[:, slot, i, 2]
            ve = FY * ce[:, 1] / ce[:, 2] + CY + ln_noise[:, slot, i, 3]
            # Line2D: A = ye - ys, B = xs - xe, C = xe*ys - xs*ye  (feature_manager.cpp:11-13)
            ln_abc[:, i] = np.stack([ve - vs, us - ue, ue * vs - us * ve], -1)
        line = dict(frame=ln_frame, pts_start=ln_ps, pts_end=ln_pe, abc=ln_abc)
    else:
        line = dict(frame=np.zeros(0, int), pts_start=np.zeros((B, 0, 3)), pts_end=np.zeros((B, 0, 3)),
                    abc=np.zeros((B, 0, 3)))

    # ---- initial state = truth + perturbation, biases zero
    pert = st_pert[:, kf_abs]                                           # (B,F,9)
    pose = np.empty((B, F, 7)); sb = np.zeros((B, F, 9))
    q_true = R2q(Rwb)
    dtheta = np.deg2rad(0.5) * pert[..., 3:6]
    dq = np.concatenate([dtheta / 2, np.ones((B, F, 1))], -1)
    pose[..., 0:3] = pwb + 0.02 * pert[..., 0:3]
    pose[..., 3:7] = qnormalize(qmul(q_true, dq))
    sb[..., 0:3] = vwb + 0.05 * pert[..., 6:9]
    ex_pose = np.broadcast_to(np.concatenate([TIC, q_ex[()]]), (B, 7)).copy()

    return dict(
        ids=ids, B=B, F=F, G=G,
        pose=pose, speedbias=sb, ex_pose=ex_pose, lam=lam0,
        imu=imu,
        proj=dict(frame_i=proj_fi, frame_j=proj_fj, landmark=proj_lm, pts_i=pts_i, pts_j=pts_j,
                  sqrt_info=PROJ_SQRT_INFO, loss_a=1.0),
        line=dict(line, K=K_MAT.copy(), Ric=Ric_n, Tic=TIC.copy(), loss_a=1.0),
        prior=None,
        truth=dict(pose=np.concatenate([pwb, q_true], -1), vel=vwb, ba=ba_true, bg=bg_true,
                   lam=lam_true, P_w=P_w, t=tk),
    )


_SHARED = {"G", "K", "Ric", "Tic", "frame_i", "frame_j", "landmark", "frame", "sqrt_info", "loss_a",
           "B", "F"}


def window_at(batch: dict, b: int) -> dict:
    """Single-window view of a batched dict (per-window arrays lose their leading axis)."""
    def take(k, v):
        if k in _SHARED or v is None:
            return v
        if isinstance(v, dict):
            return {kk: take(kk, x) for kk, x in v.items()}
        if isinstance(v, np.ndarray):
            return v[b]
        return v
    out = {k: take(k, v) for k, v in batch.items()}
    out["B"] = 1
    return out


def with_time_offset(win: dict, seed: int = 0, td_true: float = 0.004, TR: float = 0.02, ROW: float = IMG_H) -> dict:
    """ESTIMATE_TD variant of a single window (estimator.cpp:1703-1707, :1757-1763): every point factor becomes a
    ProjectionTdFactor on the extra block para_Td[0].  Feature velocities on the normalised plane, image rows and the time
    offsets the observations were stamped with (td_i = td_j = 0, the previous estimate) are drawn at random; the observations
    are shifted so that they are consistent at td = td_true (rolling-shutter read-out TR over ROW rows).  td starts at 0."""
    rng = np.random.Generator(np.random.PCG64(0x7D + seed))
    pr = dict(win["proj"])
    n = len(pr["frame_i"])
    vel_i = rng.normal(size=(n, 2)) * 0.4; vel_j = rng.normal(size=(n, 2)) * 0.4
    row_i = rng.uniform(0.0, ROW, n); row_j = rng.uniform(0.0, ROW, n)
    td_i = np.zeros(n); td_j = np.zeros(n)
    pts_i = np.array(pr["pts_i"], dtype=float).copy(); pts_j = np.array(pr["pts_j"], dtype=float).copy()
    pts_i[:, :2] += (td_true - td_i + TR / ROW * (row_i - ROW / 2))[:, None] * vel_i      # projection_td_factor.cpp:50-51 undone
    pts_j[:, :2] += (td_true - td_j + TR / ROW * (row_j - ROW / 2))[:, None] * vel_j
    pr.update(pts_i=pts_i, pts_j=pts_j, vel_i=vel_i, vel_j=vel_j, td_i=td_i, td_j=td_j, row_i=row_i, row_j=row_j, TR=float(TR), ROW=float(ROW))
    out = dict(win, proj=pr, td=0.0)
    return out
