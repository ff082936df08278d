"""Front-end-free replay harness: the callers of the hot path.

The reference's steady state is `estimator_node.cpp:process()` -> `Estimator::processIMU` (estimator.cpp:191-228) ->
`processImagewithLine` (:230-383) -> `solveOdometry` (:1476-1490: triangulate + OptimizationWithLine) ->
`failureDetection` (:1629-1675) -> `slideWindowWithLinesFoV` (:2121-2259) with `FeatureManager`
(feature_manager.cpp: addFeaturesCheckParallax :260-334, triangulate :440-492, setDepth :379-397, removeFailures :399-408,
removeBackShiftDepth :559-616, removeFront :655-696, compensatedParallax2 :698-734).  This module mirrors exactly that
window management on the host (NumPy) and hands every window to a BACK END:

    backend.preintegrate(bufs)            -> IntegrationBase results for a list of IMU buffers
    backend.optimize(win, marg_flag)      -> solved + gauge-fixed states and the new marginalisation prior

`HipBackend` is the product path (the C-ABI of include/tcv.h: device pre-integration, fused solve, gauge fix and
marginalisation kernels).  The tests plug the CPU oracle in through the same interface and compare trajectories (ATE).

The image/IMU front end (feature tracker, line detector, ROS) is out of scope; `simulate_stream` produces the per-frame
streams it would deliver -- IMU samples, tracked point features, matched 2D-3D lines -- from an analytic trajectory with the
sensor model of benchmark_publisher/config/V1_01_easy/sensor.yaml, `simulate_stream_euroc` along the EuRoC ground-truth
trajectories the reference ships (data/euroc_*.npz).  Initialisation (initialStructure, estimator.cpp:1221) is out of scope too:
the first window starts from perturbed ground truth.  `run_many_native` drives the same logic in native code
(include/tcv_estimator.h); this module stays the reference for it and the vehicle for the oracle comparison.
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np

import synth

MARGIN_OLD, MARGIN_SECOND_NEW = 0, 1
WINDOW_SIZE = synth.WINDOW_SIZE
MIN_PARALLAX = 10.0 / synth.FOCAL_LENGTH        # keyframe_parallax: 10 px (sensor.yaml:87, parameters.cpp:73-74)
INIT_DEPTH = 5.0                                # parameters.cpp:131
G = np.array([0.0, 0.0, synth.G_NORM])


# ----------------------------------------------------------------------------------------------------------------
# small rotation helpers (same conventions as the factors: quaternions x y z w)
# ----------------------------------------------------------------------------------------------------------------
def q2R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R2q(m):
    """Eigen `Quaterniond q{R}` (vector2double, estimator.cpp:1499)."""
    q = np.zeros(4)
    t = m[0, 0] + m[1, 1] + m[2, 2]
    if t > 0:
        t = np.sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t
        q[0] = (m[2, 1] - m[1, 2]) * t; q[1] = (m[0, 2] - m[2, 0]) * t; q[2] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j = (i + 1) % 3; k = (j + 1) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0); q[i] = 0.5 * t; t = 0.5 / t
        q[3] = (m[k, j] - m[j, k]) * t; q[j] = (m[j, i] + m[i, j]) * t; q[k] = (m[k, i] + m[i, k]) * t
    return q


def deltaQ_R(theta):
    """Utility::deltaQ(theta).toRotationMatrix() (utility.h:15-28; not normalised, like the reference)."""
    return q2R(np.array([theta[0] / 2, theta[1] / 2, theta[2] / 2, 1.0]))


# ----------------------------------------------------------------------------------------------------------------
# simulated front-end streams
# ----------------------------------------------------------------------------------------------------------------
def simulate_stream(seed: int, n_frames: int, max_features: int = 36, max_lines: int = 4, pixel_sigma: float = 1.0,
                    t0: float = 0.0, imu_noise: bool = True, pace: float = 0.85, associate: bool = False, hold=None):
    """Per-frame streams of a front end following the analytic trajectory of synth.py at 10 Hz / 200 Hz IMU.
    Returns dict(t, gt_p, gt_R, gt_v, imu=[(acc (S+1,3), gyr (S+1,3))] per frame (sample 0 = the previous frame's last
    sample), points=[{id: (x, y, 1)}], lines=[[(pts_start, pts_end, abc)]], ba, bg).
    associate=True: the lines come as the line tracker delivers them -- [(track id, pixel end points xs ys xe ye)] without their
    3D partner -- together with the prior map (`map_lines` n x 6 in the map frame, `Rbw`, `Tbw`); the replay then runs the
    reference's 2D-3D association (UpdateLinesInFoV / LineCorrespondenceInFrame / removeLineOutlier) every frame.
    hold=(t_stop, seconds): the platform comes to rest at t_stop (over 0.5 s), stands still for `seconds` and moves on -- every frame of the
    rest is a non-keyframe (MARGIN_SECOND_NEW), the newest interval's IMU buffers are merged frame after frame, and beyond 10 s the window's
    last IMU factor is left out (estimator.cpp:1726) and so is MARGIN_OLD's (:1933) when the rest ends."""
    rng = np.random.Generator(np.random.PCG64(0xFEED + seed))
    S = synth.IMU_RATE_SUB
    t = t0 + synth.DT_KF * np.arange(n_frames)
    ba = rng.uniform(-0.05, 0.05, 3); bg = rng.uniform(-0.01, 0.01, 3)
    ts = t0 + synth.DT_IMU * np.arange((n_frames - 1) * S + 1)
    # smooth time warp tau(t): the platform alternates between fast and almost-hovering phases, so that both keyframe
    # decisions of addFeaturesCheckParallax (MARGIN_OLD / MARGIN_SECOND_NEW) occur
    wa, ww = pace, 2.0 * np.pi / 3.0
    tau0 = lambda x: x + wa / ww * np.sin(ww * x)
    dtau0 = lambda x: 1.0 + wa * np.cos(ww * x)
    ddtau0 = lambda x: -wa * ww * np.sin(ww * x)
    if hold is None:
        tau, dtau, ddtau = tau0, dtau0, ddtau0
    else:
        # clock of the platform c(x): runs with the wall clock, slows to a stop over `w` seconds at t_stop (C2 smoothstep), rests, speeds up again
        ts0, rest = float(hold[0]), float(hold[1])
        w = 0.5
        sm = lambda u: ((6.0 * u - 15.0) * u + 10.0) * u ** 3                       # smoothstep and its integral / derivative
        ism = lambda u: ((u - 3.0) * u + 2.5) * u ** 4
        dsm = lambda u: ((30.0 * u - 60.0) * u + 30.0) * u ** 2

        def bump(x):      # 1 while at rest, 0 while moving; (integral from -inf, value, derivative)
            x = np.asarray(x, dtype=float)
            a, b = ts0, ts0 + w + rest
            u1 = np.clip((x - a) / w, 0.0, 1.0); u2 = np.clip((x - b) / w, 0.0, 1.0)
            val = sm(u1) - sm(u2)
            integ = w * ism(u1) + np.clip(x - (a + w), 0.0, None) - (w * ism(u2) + np.clip(x - (b + w), 0.0, None))
            der = (dsm(u1) * ((x > a) & (x < a + w)) - dsm(u2) * ((x > b) & (x < b + w))) / w
            return integ, val, der
        cl = lambda x: np.asarray(x, dtype=float) - bump(x)[0]
        tau = lambda x: tau0(cl(x))
        dtau = lambda x: dtau0(cl(x)) * (1.0 - bump(x)[1])
        ddtau = lambda x: ddtau0(cl(x)) * (1.0 - bump(x)[1]) ** 2 - dtau0(cl(x)) * bump(x)[2]
    Rw = synth.traj_R(tau(ts))
    a_w = synth.traj_a(tau(ts)) * dtau(ts)[:, None] ** 2 + synth.traj_v(tau(ts)) * ddtau(ts)[:, None]
    acc = (np.swapaxes(Rw, -1, -2) @ (a_w + G)[..., None])[..., 0] + ba
    gyr = synth.traj_w_body(tau(ts)) * dtau(ts)[:, None] + bg
    if imu_noise:      # discrete-time white noise of a 200 Hz IMU with the continuous densities of sensor.yaml:90-91
        acc = acc + rng.normal(size=acc.shape) * synth.ACC_N * 0.1
        gyr = gyr + rng.normal(size=gyr.shape) * synth.GYR_N * 0.1
    imu = [None] + [(acc[(k - 1) * S:k * S + 1].copy(), gyr[(k - 1) * S:k * S + 1].copy()) for k in range(1, n_frames)]
    Rk = synth.traj_R(tau(t)); pk = synth.traj_p(tau(t)); vk = synth.traj_v(tau(t)) * dtau(t)[:, None]
    ps_pool, pe_pool = synth.line_pool()
    points, lines = _front_end(rng, Rk, pk, ps_pool, pe_pool, max_features, max_lines, pixel_sigma, associate)
    out = dict(t=t, gt_p=pk, gt_R=Rk, gt_v=vk, imu=imu, points=points, lines=lines, ba=ba, bg=bg)
    if associate:
        out.update(map_lines=np.hstack([(ps_pool - synth.TBW) @ synth.RBW, (pe_pool - synth.TBW) @ synth.RBW]), Rbw=synth.RBW.copy(), Tbw=synth.TBW.copy())
    return out


def _front_end(rng, Rk, pk, ps_pool, pe_pool, max_features, max_lines, pixel_sigma, associate):
    """what the feature tracker and the line tracker would deliver along the body poses (Rk, pk): per frame {id: (x, y, 1)}
    tracked points on the normalised plane (a lost track is never re-acquired, new landmarks are spawned in front of the camera
    when the tracker runs short) and the visible lines of the pool (world-frame end points ps_pool / pe_pool)."""
    n_frames = len(pk)
    Rwc, twc = synth._cam_pose(Rk, pk)
    land = []            # world points, spawned in front of the camera when the tracker runs short of features
    alive = []
    points, lines = [], []
    for k in range(n_frames):
        obs = {}
        if land:
            Pw = np.array(land)
            pc = synth._project(Rwc[k], twc[k], Pw)
            vis = synth._visible_norm(pc, 8.0) & (pc[:, 2] < 14.0)
            for i in np.nonzero(vis & np.array(alive))[0]:
                obs[int(i)] = pc[i]
            for i in range(len(land)):
                if alive[i] and not vis[i]:
                    alive[i] = False          # a lost track is never re-acquired (feature_tracker behaviour)
        while len(obs) < max_features:
            d = rng.uniform(3.0, 8.0)
            u = rng.uniform(30.0, synth.IMG_W - 30.0); v = rng.uniform(30.0, synth.IMG_H - 30.0)
            pc = np.array([(u - synth.CX) / synth.FX * d, (v - synth.CY) / synth.FY * d, d])
            land.append(Rwc[k] @ pc + twc[k]); alive.append(True)
            obs[len(land) - 1] = pc
        frame_pts = {}
        for i in sorted(obs)[:max_features]:
            pc = obs[i]
            n = rng.normal(size=2) * pixel_sigma / synth.FOCAL_LENGTH
            frame_pts[i] = np.array([pc[0] / pc[2] + n[0], pc[1] / pc[2] + n[1], 1.0])
        points.append(frame_pts)
        fl = []
        pcs = synth._project(Rwc[k], twc[k], ps_pool); pce = synth._project(Rwc[k], twc[k], pe_pool)
        ok = np.nonzero(synth._visible_norm(pcs, 5.0) & synth._visible_norm(pce, 5.0))[0]
        for i in ok[:max_lines]:
            uv = []
            for pc in (pcs[i], pce[i]):
                uv.append(np.array([synth.FX * pc[0] / pc[2] + synth.CX, synth.FY * pc[1] / pc[2] + synth.CY]) + rng.normal(size=2) * pixel_sigma)
            (xs, ys), (xe, ye) = uv
            abc = np.array([ye - ys, xs - xe, xe * ys - xs * ye])          # feature_manager.cpp:11-13
            fl.append((int(i), np.array([xs, ys, xe, ye])) if associate else (ps_pool[i].copy(), pe_pool[i].copy(), abc))
        lines.append(fl)
    return points, lines


EUROC_SEQUENCES = ("V1_02_medium", "V1_03_difficult", "V2_01_easy", "V2_02_medium", "V2_03_difficult")


def load_euroc(seq: str):
    """the excerpt of benchmark_publisher/config/<seq>/ shipped in data/ (tests/golden/make_euroc_excerpts.py): ground-truth
    states at 200 Hz in the EuRoC reference frame (p, q wxyz, v, bw, ba: the columns benchmark_publisher_node.cpp:42-62 parses),
    the prior 3D line map (map frame = ground-truth frame) and initialRotation / initialTranslation (sensor.yaml:40-57), the
    transform Rbw, Tbw from that frame into the VIO world frame."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "euroc_%s.npz" % seq))
    st = np.cumsum(d["state_micro_delta"].astype(np.int64), axis=0) * 1e-6
    for key, ref in (("Ric", synth.RIC), ("Tic", synth.TIC), ("K", np.array([synth.FX, synth.FY, synth.CX, synth.CY])), ("size", np.array([synth.IMG_W, synth.IMG_H])),
                     ("imu_noise", np.array([synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W, synth.G_NORM]))):
        if not np.array_equal(d[key], ref):
            raise ValueError("euroc_%s: %s differs from the calibration the kernels' host side is configured with" % (seq, key))
    return dict(seq=seq, stamp_ns=d["stamp_ns"], t=(d["stamp_ns"] - d["stamp_ns"][0]) * 1e-9, p=st[:, 0:3], q_wxyz=st[:, 3:7], v=st[:, 7:10], bw=st[:, 10:13],
                ba=st[:, 13:16], lines3d=d["lines3d"], Rbw=d["Rbw"], Tbw=d["Tbw"], line_th=d["line_th"])


def _rotvec(R):
    """rotation vectors of a stack of rotation matrices (angles well below pi)."""
    w = np.stack([R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]], axis=-1) * 0.5
    s = np.linalg.norm(w, axis=-1)
    c = (np.trace(R, axis1=-2, axis2=-1) - 1.0) * 0.5
    ang = np.arctan2(s, c)
    return w * np.where(s > 1e-12, ang / np.where(s > 1e-12, s, 1.0), 1.0)[..., None]


def simulate_stream_euroc(seq, n_frames: int, start_s: float = 0.0, seed: int = 0, max_features: int = 36, max_lines: int = 4,
                          pixel_sigma: float = 1.0, imu_noise: bool = True, associate: bool = False):
    """Per-frame streams (same dict as `simulate_stream`) of a front end carried along the EuRoC ground-truth trajectory of
    `seq` (name or a `load_euroc` dict), starting `start_s` seconds into the excerpt: SURVEY.md 8(f) N1's "simulator producing
    those streams from GT data.csv + line_3d.txt + sensor.yaml".  The bag (images, raw IMU) is not part of the reference, so
      * the trajectory is the ground truth mapped into the VIO world frame, x_w = Rbw x + Tbw (gravity along -z of that frame);
      * the 200 Hz IMU samples are what an ideal IMU on that trajectory reads -- body rates from consecutive ground-truth
        attitudes, specific force from the central difference of the ground-truth velocity -- plus the ground truth's own
        (slowly varying) gyroscope / accelerometer biases and white noise of the densities in sensor.yaml:90-91;
      * points are tracked landmarks spawned in view, lines are the sequence's own prior map (891 / 908 segments) seen from the
        ground-truth camera, with pixel noise -- `associate=True` leaves the 2D-3D association to the replay."""
    E = load_euroc(seq) if isinstance(seq, str) else seq
    rng = np.random.Generator(np.random.PCG64(0xE0C0 + seed))
    S = synth.IMU_RATE_SUB
    i0 = int(round(start_s / synth.DT_IMU)) + 1                 # one row of margin on both sides for the central differences
    n_s = (n_frames - 1) * S + 1
    if i0 + n_s + 1 > len(E["t"]):
        raise ValueError("euroc excerpt too short: %d frames from %.1f s need %d rows, the excerpt holds %d" % (n_frames, start_s, i0 + n_s + 1, len(E["t"])))
    U, _, Vt = np.linalg.svd(E["Rbw"])
    Rbw = U @ Vt                                                 # the 6-digit matrix of the yaml, re-orthonormalised
    sl = slice(i0 - 1, i0 + n_s + 1)
    q = E["q_wxyz"][sl] / np.linalg.norm(E["q_wxyz"][sl], axis=1)[:, None]
    Rg = np.array([q2R(np.array([a[1], a[2], a[3], a[0]])) for a in q])
    Rw = Rbw @ Rg
    pw = E["p"][sl] @ Rbw.T + E["Tbw"]
    vw = E["v"][sl] @ Rbw.T
    dt = synth.DT_IMU
    w_half = _rotvec(np.swapaxes(Rw[:-1], -1, -2) @ Rw[1:]) / dt         # mean body rate over [k, k+1]
    gyr = 0.5 * (w_half[:-1] + w_half[1:]) + E["bw"][sl][1:-1]
    a_w = (vw[2:] - vw[:-2]) / (2.0 * dt)
    Rw, pw, vw = Rw[1:-1], pw[1:-1], vw[1:-1]
    acc = (np.swapaxes(Rw, -1, -2) @ (a_w + G)[..., None])[..., 0] + E["ba"][sl][1:-1]
    if imu_noise:
        acc = acc + rng.normal(size=acc.shape) * synth.ACC_N * 0.1
        gyr = gyr + rng.normal(size=gyr.shape) * synth.GYR_N * 0.1
    imu = [None] + [(acc[(k - 1) * S:k * S + 1].copy(), gyr[(k - 1) * S:k * S + 1].copy()) for k in range(1, n_frames)]
    fr = np.arange(n_frames) * S
    ps_pool = E["lines3d"][:, 0:3] @ E["Rbw"].T + E["Tbw"]        # estimator.cpp:1832-1833 (the yaml's matrix as it is)
    pe_pool = E["lines3d"][:, 3:6] @ E["Rbw"].T + E["Tbw"]
    points, lines = _front_end(rng, Rw[fr], pw[fr], ps_pool, pe_pool, max_features, max_lines, pixel_sigma, associate)
    out = dict(t=E["t"][sl][1:-1][fr], stamp_ns=E["stamp_ns"][sl][1:-1][fr], gt_p=pw[fr], gt_R=Rw[fr], gt_v=vw[fr], imu=imu, points=points, lines=lines,
               ba=E["ba"][sl][1 + fr[0]].copy(), bg=E["bw"][sl][1 + fr[0]].copy(), seq=E["seq"])
    if associate:
        out.update(map_lines=E["lines3d"].copy(), Rbw=E["Rbw"].copy(), Tbw=E["Tbw"].copy(), line_th=tuple(float(v) for v in E["line_th"]))      # the sequence's own angle_th, overlap_th, dist_th
    return out


# ----------------------------------------------------------------------------------------------------------------
# window management (Estimator + FeatureManager mirror)
# ----------------------------------------------------------------------------------------------------------------
class Feature:
    __slots__ = ("id", "start_frame", "obs", "depth", "solve_flag")

    def __init__(self, fid, start_frame):
        self.id, self.start_frame, self.obs, self.depth, self.solve_flag = fid, start_frame, [], -1.0, 0      # FeaturePerId ctor: estimated_depth(-1.0)

    def end_frame(self):
        return self.start_frame + len(self.obs) - 1


class Replay:
    def __init__(self, backend, num_iterations: int = 8, fixed_iterations: bool = False):
        W = WINDOW_SIZE
        self.backend = backend
        self.num_iterations, self.fixed_iterations = num_iterations, fixed_iterations
        self.Ps = np.zeros((W + 1, 3)); self.Rs = np.tile(np.eye(3), (W + 1, 1, 1)); self.Vs = np.zeros((W + 1, 3))
        self.Bas = np.zeros((W + 1, 3)); self.Bgs = np.zeros((W + 1, 3))
        self.tic = synth.TIC.copy(); self.ric = synth.RIC.copy()
        self.bufs = [None] * (W + 1)       # per slot: dict(acc0, gyr0, ba, bg, acc [S,3], gyr [S,3]) = pre_integrations[j] + dt/acc/gyr_buf[j]
        self.pre = [None] * (W + 1)        # cached pre-integration of every slot
        self.features = []                 # f_manager.feature (insertion order matters: it is the landmark order)
        self.line_obs = [[] for _ in range(W + 1)]
        self.map_lines = None              # prior 3D line map (association mode): lines3d_map, Rbw, Tbw
        self.linefeatures = []             # f_manager.linefeature: dict(id, start_frame, obs=[...], credible_matching)
        self.fov = [None] * (W + 1)        # WorldLinesInFOV[i] as a mask over the map
        self.fov_ready = False
        self.exact_line_jacobian = False        # True: tcv_problem_set_line_jacobian(p, 1) (opt-in extension, not the reference's behaviour)
        self.angle_th, self.overlap_th, self.dist_th = 0.1745, 0.45, 50.0      # sensor.yaml (V1_01_easy): angle_th, overlap_th, dist_th (estimator.cpp:116-119)
        self.prior = None                  # last_marginalization_info + last_marginalization_parameter_blocks
        self.frame_count = 0
        self.marg_flag = MARGIN_OLD
        self.acc_0 = None; self.gyr_0 = None
        self.last_P = None; self.last_R = None
        self.log = []

    # ---- IMU ---------------------------------------------------------------------------------------------------
    def process_imu(self, acc, gyr):
        """Estimator::processIMU for all samples between two frames; acc/gyr (S+1, 3), row 0 = the previous frame's last sample."""
        j = self.frame_count
        if self.acc_0 is None:
            self.acc_0, self.gyr_0 = acc[0].copy(), gyr[0].copy()
        if self.bufs[j] is None:
            self.bufs[j] = dict(acc0=self.acc_0.copy(), gyr0=self.gyr_0.copy(), ba=self.Bas[j].copy(), bg=self.Bgs[j].copy(), acc=[], gyr=[])
        dt = synth.DT_IMU
        for a, w in zip(acc[1:], gyr[1:]):
            if j != 0:
                self.bufs[j]["acc"].append(a.copy()); self.bufs[j]["gyr"].append(w.copy())
                un_acc_0 = self.Rs[j] @ (self.acc_0 - self.Bas[j]) - G               # estimator.cpp:217-224
                un_gyr = 0.5 * (self.gyr_0 + w) - self.Bgs[j]
                self.Rs[j] = self.Rs[j] @ deltaQ_R(un_gyr * dt)
                un_acc_1 = self.Rs[j] @ (a - self.Bas[j]) - G
                un_acc = 0.5 * (un_acc_0 + un_acc_1)
                self.Ps[j] = self.Ps[j] + dt * self.Vs[j] + 0.5 * dt * dt * un_acc
                self.Vs[j] = self.Vs[j] + dt * un_acc
            self.acc_0, self.gyr_0 = a.copy(), w.copy()
        self.pre[j] = None

    # ---- FeatureManager ------------------------------------------------------------------------------------------
    def add_features_check_parallax(self, pts: dict, lines: list) -> bool:
        fc = self.frame_count
        by_id = {f.id: f for f in self.features}
        last_track_num = 0
        for fid, p in pts.items():
            f = by_id.get(fid)
            if f is None:
                f = Feature(fid, fc); self.features.append(f)
            else:
                last_track_num += 1
            f.obs.append(np.asarray(p, dtype=float))
        if self.map_lines is not None:       # the line tracker's (id, end points): addFeaturesCheckParallax :291-311
            by_lid = {lf["id"]: lf for lf in self.linefeatures}
            for lid, v in lines:
                lf = by_lid.get(lid)
                if lf is None:
                    lf = dict(id=lid, start_frame=fc, obs=[], credible_matching=True); self.linefeatures.append(lf)
                v = np.asarray(v, dtype=float)
                lf["obs"].append(dict(vec=v, abc=np.array([v[3] - v[1], v[0] - v[2], v[2] * v[1] - v[0] * v[3]]), world=np.zeros(6), errA=-1.0, errD=-1.0,
                                      overlap=-1.0, credible_line=True, use_flag=False))
        else:
            self.line_obs[fc] = list(lines)
        if fc < 2 or last_track_num < 20:
            return True
        s, n = 0.0, 0
        for f in self.features:
            if f.start_frame <= fc - 2 and f.end_frame() >= fc - 1:
                pi, pj = f.obs[fc - 2 - f.start_frame], f.obs[fc - 1 - f.start_frame]      # compensatedParallax2 (:698-734)
                du, dv = pi[0] / pi[2] - pj[0], pi[1] / pi[2] - pj[1]
                s += np.sqrt(du * du + dv * dv); n += 1
        return True if n == 0 else (s / n >= MIN_PARALLAX)

    def _selected(self):
        return [f for f in self.features if len(f.obs) >= 2 and f.start_frame < WINDOW_SIZE - 2]

    def triangulate(self):
        for f in self._selected():
            if f.depth > 0:
                continue
            i = f.start_frame
            t0 = self.Ps[i] + self.Rs[i] @ self.tic; R0 = self.Rs[i] @ self.ric
            A = []
            for k, p in enumerate(f.obs):
                j = i + k
                t1 = self.Ps[j] + self.Rs[j] @ self.tic; R1 = self.Rs[j] @ self.ric
                t = R0.T @ (t1 - t0); R = R0.T @ R1
                P = np.hstack([R.T, (-R.T @ t)[:, None]])
                fn = p / np.linalg.norm(p)
                A.append(fn[0] * P[2] - fn[2] * P[0]); A.append(fn[1] * P[2] - fn[2] * P[1])
            V = np.linalg.svd(np.array(A))[2][-1]
            f.depth = V[2] / V[3]
            if f.depth < 0.1:
                f.depth = INIT_DEPTH

    # ---- window -> back end ----------------------------------------------------------------------------------------
    def _preintegrations(self):
        need = [j for j in range(1, WINDOW_SIZE + 1) if self.pre[j] is None]
        if need:
            res = self.backend.preintegrate([self.bufs[j] for j in need])
            for j, r in zip(need, res):
                self.pre[j] = r
        return [self.pre[j] for j in range(1, WINDOW_SIZE + 1)]

    def _poses(self):
        W = WINDOW_SIZE
        pose = np.zeros((W + 1, 7))
        for i in range(W + 1):
            pose[i, :3] = self.Ps[i]; pose[i, 3:] = R2q(self.Rs[i])
        return pose, np.concatenate([self.tic, R2q(self.ric)])

    def associate_lines(self):
        """UpdateLinesInFoV(frame_count) (or initialLineFoVWindow the first time), updateLinePairInWindow (estimator.cpp:449-481)
        and f_manager.removeLineOutlier (feature_manager.cpp:494-534), as processImagewithLine runs them before solveOdometry."""
        W = WINDOW_SIZE
        pose, ex = self._poses()
        det_frame, det, where = [], [], []
        for lf in self.linefeatures:
            for k, ob in enumerate(lf["obs"]):
                det_frame.append(lf["start_frame"] + k); det.append(ob["vec"]); where.append(ob)
        given = None
        if self.fov_ready:
            given = np.array([self.fov[i] if self.fov[i] is not None else np.zeros(len(self.map_lines), bool) for i in range(W + 1)])
        fov_now, _, _, _ = self.backend.match_lines(pose, ex, None, np.zeros(0, np.int32), np.zeros((0, 4)), map3d=(self.map_lines, self.Rbw, self.Tbw), th=(self.angle_th, self.overlap_th))
        if not self.fov_ready:
            for i in range(W + 1):
                self.fov[i] = fov_now[i].copy()          # initialLineFoVWindow (:483-497)
            self.fov_ready = True
        else:
            self.fov[W] = fov_now[W].copy()              # UpdateLinesInFoV(frame_count) (:385-447)
        given = np.array(self.fov)
        if det:
            _, match, err, _ = self.backend.match_lines(pose, ex, given, np.array(det_frame, dtype=np.int32), np.array(det), map3d=(self.map_lines, self.Rbw, self.Tbw), th=(self.angle_th, self.overlap_th))
            for ob, f, m, e in zip(where, det_frame, match, err):
                ob["errA"], ob["errD"], ob["overlap"] = float(e[0]), float(e[1]), float(e[2])
                idx = np.nonzero(given[f])[0]
                if m >= 0:
                    ob["world"] = self.map_lines[m].copy()
                elif len(idx):
                    ob["world"] = self.map_lines[idx[0]].copy()           # `linesInThisFov[0]` (:871-877)
                else:
                    ob["world"] = np.concatenate([[ob["vec"][0], ob["vec"][1], 1.0]] * 2)       # fake_line (:707-709)
                ob["use_flag"] = True
                ob["credible_line"] = e[0] != -1
        for lf in self.linefeatures:                      # removeLineOutlier
            if len(lf["obs"]) < 1:
                continue
            first = lf["obs"][0]["world"]; count = 0
            for ob in lf["obs"]:
                d = (ob["world"][3:] - ob["world"][:3]) - (first[3:] - first[:3])
                ob["credible_line"] = not (np.float32(np.linalg.norm(d)) > 0.1)
                count += 0 if ob["credible_line"] else 1
            lf["credible_matching"] = not ((count // len(lf["obs"])) >= 0.5)

    def _line_factors(self):
        """estimator.cpp:1786-1846: which line observations become LineProjectionFactors."""
        lf_, lps, lpe, labc = [], [], [], []
        if self.map_lines is None:
            for i in range(WINDOW_SIZE + 1):
                for (ps, pe, abc) in self.line_obs[i]:
                    lf_.append(i); lps.append(ps); lpe.append(pe); labc.append(abc)
            return lf_, lps, lpe, labc
        dist_th = self.dist_th
        for lf in self.linefeatures:
            if not (len(lf["obs"]) >= 2 and lf["start_frame"] < WINDOW_SIZE - 2) or not lf["credible_matching"]:
                continue
            for k, ob in enumerate(lf["obs"]):
                if not ob["credible_line"] or not ob["use_flag"] or ob["errD"] > dist_th:
                    continue
                lf_.append(lf["start_frame"] + k)
                lps.append(self.Rbw @ ob["world"][:3] + self.Tbw); lpe.append(self.Rbw @ ob["world"][3:] + self.Tbw); labc.append(ob["abc"])
        return lf_, lps, lpe, labc

    def build_window(self):
        W = WINDOW_SIZE
        pose = np.zeros((W + 1, 7)); sb = np.zeros((W + 1, 9))
        for i in range(W + 1):      # vector2double (estimator.cpp:1492-1535)
            pose[i, :3] = self.Ps[i]; pose[i, 3:] = R2q(self.Rs[i])
            sb[i, :3] = self.Vs[i]; sb[i, 3:6] = self.Bas[i]; sb[i, 6:] = self.Bgs[i]
        ex = np.concatenate([self.tic, R2q(self.ric)])
        sel = self._selected()
        lam = np.array([1.0 / f.depth for f in sel])
        pre = self._preintegrations()
        keep = [k for k in range(W) if pre[k]["sum_dt"] <= 10.0]          # estimator.cpp:1726
        imu = {key: np.array([pre[k][key] for k in keep]) for key in ("delta_p", "delta_q", "delta_v", "lin_ba", "lin_bg", "sum_dt", "jacobian", "covariance")}
        imu["frame_i"] = np.array(keep, dtype=np.int64); imu["frame_j"] = np.array(keep, dtype=np.int64) + 1
        fi, fj, fl, pi, pj = [], [], [], [], []
        for l, f in enumerate(sel):      # estimator.cpp:1737-1771
            for k in range(1, len(f.obs)):
                fi.append(f.start_frame); fj.append(f.start_frame + k); fl.append(l); pi.append(f.obs[0]); pj.append(f.obs[k])
        proj = dict(frame_i=np.array(fi, dtype=np.int64), frame_j=np.array(fj, dtype=np.int64), landmark=np.array(fl, dtype=np.int64),
                    pts_i=np.array(pi).reshape(-1, 3), pts_j=np.array(pj).reshape(-1, 3), sqrt_info=synth.PROJ_SQRT_INFO, loss_a=1.0)
        lf, lps, lpe, labc = self._line_factors()
        line = dict(frame=np.array(lf, dtype=np.int64), pts_start=np.array(lps).reshape(-1, 3), pts_end=np.array(lpe).reshape(-1, 3),
                    abc=np.array(labc).reshape(-1, 3), K=synth.K_MAT.copy(), Ric=q2R(ex[3:] / np.linalg.norm(ex[3:])), Tic=self.tic.copy(), loss_a=1.0,
                    exact_jacobian=self.exact_line_jacobian)
        win = dict(pose=pose, speedbias=sb, ex_pose=ex, lam=lam, imu=imu, proj=proj, line=line, G=G.copy(), prior=self.prior)
        return win, sel

    def prepare_window(self):
        """solveOdometry (:1476-1490) up to the solver call: association, triangulation, vector2double + graph."""
        if self.map_lines is not None:
            self.associate_lines()
        self.triangulate()
        self._win, self._sel = self.build_window()
        return self._win

    def apply_result(self, out):
        """double2vector (:1565-1581; the gauge fix itself runs in the back end), setDepth, the new prior."""
        win, sel = self._win, self._sel
        W = WINDOW_SIZE
        for i in range(W + 1):
            self.Ps[i] = out["pose"][i, :3]; self.Rs[i] = q2R(out["pose"][i, 3:])
            self.Vs[i] = out["sb"][i, :3]; self.Bas[i] = out["sb"][i, 3:6]; self.Bgs[i] = out["sb"][i, 6:]
        self.tic = out["ex"][:3].copy(); self.ric = q2R(out["ex"][3:])
        for f, lam in zip(sel, out["lam"]):      # setDepth (feature_manager.cpp:379-397)
            f.depth = 1.0 / lam
            f.solve_flag = 2 if f.depth < 0 else 1
        if out.get("prior", "keep") != "keep":
            self.prior = out["prior"]
        self.log.append(dict(flag=self.marg_flag, n_landmarks=len(sel), n_proj=len(win["proj"]["frame_i"]), n_line=len(win["line"]["frame"]), n_line_obs=sum(len(lf["obs"]) for lf in self.linefeatures),
                             iterations=out.get("iterations"), final_cost=out.get("final_cost"), prior_n=None if self.prior is None else self.prior["n"]))

    def optimize(self):
        """solveOdometry + double2vector + the marginalisation of OptimizationWithLine, one window."""
        win = self.prepare_window()
        self.apply_result(self.backend.optimize(win, self.marg_flag, self.num_iterations, self.fixed_iterations))

    def failure_detection(self) -> bool:
        W = WINDOW_SIZE
        if np.linalg.norm(self.Bas[W]) > 2.5 or np.linalg.norm(self.Bgs[W]) > 1.0:
            return True
        if self.last_P is not None:
            if np.linalg.norm(self.Ps[W] - self.last_P) > 5 or abs(self.Ps[W][2] - self.last_P[2]) > 1:
                return True
        return False

    # ---- sliding ---------------------------------------------------------------------------------------------------
    def slide_window(self):
        W = WINDOW_SIZE
        if self.frame_count != W:
            return
        if self.marg_flag == MARGIN_OLD:
            back_R0, back_P0 = self.Rs[0].copy(), self.Ps[0].copy()
            for arr in (self.Ps, self.Rs, self.Vs, self.Bas, self.Bgs):
                arr[:-1] = arr[1:].copy()            # the swaps of :2131-2153 followed by the copy of slot W-1 into W
            self.bufs = self.bufs[1:] + [None]
            self.pre = self.pre[1:] + [None]
            self.line_obs = self.line_obs[1:] + [list(self.line_obs[W])]      # WorldLinesInFOV[W] = WorldLinesInFOV[W-1] (:2158)
            self.bufs[W] = dict(acc0=self.acc_0.copy(), gyr0=self.gyr_0.copy(), ba=self.Bas[W].copy(), bg=self.Bgs[W].copy(), acc=[], gyr=[])
            # slideWindowOld (:2242-2259) -> removeBackShiftDepth (feature_manager.cpp:559-616)
            R0 = back_R0 @ self.ric; R1 = self.Rs[0] @ self.ric
            P0 = back_P0 + back_R0 @ self.tic; P1 = self.Ps[0] + self.Rs[0] @ self.tic
            kept = []
            for f in self.features:
                if f.start_frame != 0:
                    f.start_frame -= 1; kept.append(f); continue
                uv_i = f.obs.pop(0)
                if len(f.obs) < 2:
                    continue
                pts_j = R1.T @ (R0 @ (uv_i * f.depth) + P0 - P1)
                f.depth = pts_j[2] if pts_j[2] > 0 else INIT_DEPTH
                kept.append(f)
            self.features = kept
            keptl = []                                # line features: feature_manager.cpp:598-614
            for lf in self.linefeatures:
                if lf["start_frame"] != 0:
                    lf["start_frame"] -= 1; keptl.append(lf); continue
                lf["obs"].pop(0)
                if len(lf["obs"]) > 0:
                    keptl.append(lf)
            self.linefeatures = keptl
            self.fov = self.fov[1:] + [self.fov[W]]
        else:
            # MARGIN_SECOND_NEW (:2189-2230): the newest frame replaces the second newest, their IMU buffers are concatenated
            self.bufs[W - 1]["acc"] += self.bufs[W]["acc"]; self.bufs[W - 1]["gyr"] += self.bufs[W]["gyr"]
            self.pre[W - 1] = None
            for arr in (self.Ps, self.Rs, self.Vs, self.Bas, self.Bgs):
                arr[W - 1] = arr[W].copy()
            self.line_obs[W - 1] = list(self.line_obs[W])
            self.bufs[W] = dict(acc0=self.acc_0.copy(), gyr0=self.gyr_0.copy(), ba=self.Bas[W].copy(), bg=self.Bgs[W].copy(), acc=[], gyr=[])
            self.pre[W] = None
            kept = []                                  # slideWindowNew -> removeFront(frame_count) (feature_manager.cpp:655-675)
            for f in self.features:
                if f.start_frame == W:
                    f.start_frame -= 1; kept.append(f); continue
                if f.end_frame() < W - 1:
                    kept.append(f); continue
                f.obs.pop(W - 1 - f.start_frame)
                if len(f.obs) > 0:
                    kept.append(f)
            self.features = kept
            keptl = []                                # feature_manager.cpp:677-695
            for lf in self.linefeatures:
                if lf["start_frame"] == W:
                    lf["start_frame"] -= 1; keptl.append(lf); continue
                if lf["start_frame"] + len(lf["obs"]) - 1 < W - 1:
                    keptl.append(lf); continue
                lf["obs"].pop(W - 1 - lf["start_frame"])
                if len(lf["obs"]) > 0:
                    keptl.append(lf)
            self.linefeatures = keptl
            self.fov[W - 1] = self.fov[W]
        self.features = [f for f in self.features if f.solve_flag != 2]          # removeFailures (:399-408)

    # ---- one frame ---------------------------------------------------------------------------------------------------
    def begin_frame(self, imu, pts, lines, truth=None) -> bool:
        """processIMU for the interval + the first half of processImagewithLine; True if the window is full and must be optimised.
        While the window fills (solver_flag == INITIAL) the states come from `truth` = (P, R, V): initialisation is out of scope."""
        W = WINDOW_SIZE
        if imu is not None:
            self.process_imu(*imu)
        self.marg_flag = MARGIN_OLD if self.add_features_check_parallax(pts, lines) else MARGIN_SECOND_NEW
        if truth is not None:
            self.Ps[self.frame_count], self.Rs[self.frame_count], self.Vs[self.frame_count] = truth
        if self.frame_count < W:
            self.frame_count += 1
            j = self.frame_count
            self.Bas[j] = self.Bas[j - 1]; self.Bgs[j] = self.Bgs[j - 1]
            self.Ps[j] = self.Ps[j - 1]; self.Rs[j] = self.Rs[j - 1]; self.Vs[j] = self.Vs[j - 1]
            return False
        return True

    def finish_frame(self):
        """failureDetection, the published state (pubOdometry) and slideWindow, after the optimisation."""
        W = WINDOW_SIZE
        if self.failure_detection():
            raise RuntimeError("failure detection (estimator.cpp:1629-1675): the replay diverged")
        res = (self.Ps[W].copy(), R2q(self.Rs[W]), self.Vs[W].copy())
        self.slide_window()
        self.last_P, self.last_R = self.Ps[W].copy(), self.Rs[W].copy()
        return res

    def process_frame(self, imu, pts, lines, truth=None):
        if not self.begin_frame(imu, pts, lines, truth):
            return None
        self.optimize()
        return self.finish_frame()


def run(stream: dict, backend, num_iterations: int = 8, fixed_iterations: bool = False, init_sigma=(0.02, 0.005, 0.05), bias_sigma=(0.005, 0.0005),
        exact_line_jacobian: bool = False):
    """replays a simulated stream; returns dict(t, p, q, v) of Ps/Rs/Vs[WINDOW_SIZE] after every optimisation (what pubOdometry
    writes, visualization.cpp:210-226) plus the per-frame log."""
    rng = np.random.Generator(np.random.PCG64(0xABCD))
    rp = Replay(backend, num_iterations, fixed_iterations)
    rp.exact_line_jacobian = exact_line_jacobian
    if "map_lines" in stream:
        rp.map_lines, rp.Rbw, rp.Tbw = stream["map_lines"], stream["Rbw"], stream["Tbw"]
        rp.angle_th, rp.overlap_th, rp.dist_th = stream.get("line_th", (rp.angle_th, rp.overlap_th, rp.dist_th))
        backend.set_map(stream["map_lines"], stream["Rbw"], stream["Tbw"])
    # the reference's initialisation calibrates the gyroscope bias (initial_aligment.cpp) before the first window; here the
    # biases start near the truth like the other states
    rp.Bas[:] = stream["ba"] + rng.normal(size=3) * bias_sigma[0]; rp.Bgs[:] = stream["bg"] + rng.normal(size=3) * bias_sigma[1]
    out_t, out_p, out_q, out_v = [], [], [], []
    for k in range(len(stream["t"])):
        truth = None
        if k <= WINDOW_SIZE:       # window fill: perturbed ground truth instead of initialStructure
            dth = rng.normal(size=3) * init_sigma[1]
            truth = (stream["gt_p"][k] + rng.normal(size=3) * init_sigma[0], stream["gt_R"][k] @ deltaQ_R(dth),
                     stream["gt_v"][k] + rng.normal(size=3) * init_sigma[2])
        r = rp.process_frame(stream["imu"][k], stream["points"][k], stream["lines"][k], truth)
        if r is not None:
            out_t.append(stream["t"][k]); out_p.append(r[0]); out_q.append(r[1]); out_v.append(r[2])
    return dict(t=np.array(out_t), p=np.array(out_p), q=np.array(out_q), v=np.array(out_v), log=rp.log)


def run_many(streams, backend, num_iterations: int = 8, fixed_iterations: bool = False, init_sigma=(0.02, 0.005, 0.05), bias_sigma=(0.005, 0.0005),
             exact_line_jacobian: bool = False):
    """several independent sequences in lock step (BASELINE configs[4] in spirit: per-sequence replay, sequences sharded over the
    GPUs, many per GPU): every frame the full windows of ALL sequences go to the back end as ONE batch (`optimize_many`), and the
    IMU buffers of all sequences are pre-integrated in one call.  Identical per-sequence results to `run` (the kernels do not
    couple windows).  Returns one result dict per stream."""
    S = len(streams)
    reps, rngs, outs = [], [], []
    for st in streams:
        rng = np.random.Generator(np.random.PCG64(0xABCD))
        rp = Replay(backend, num_iterations, fixed_iterations)
        rp.exact_line_jacobian = exact_line_jacobian
        if "map_lines" in st:
            rp.map_lines, rp.Rbw, rp.Tbw = st["map_lines"], st["Rbw"], st["Tbw"]
            rp.angle_th, rp.overlap_th, rp.dist_th = st.get("line_th", (rp.angle_th, rp.overlap_th, rp.dist_th))
            backend.set_map(st["map_lines"], st["Rbw"], st["Tbw"])
        rp.Bas[:] = st["ba"] + rng.normal(size=3) * bias_sigma[0]; rp.Bgs[:] = st["bg"] + rng.normal(size=3) * bias_sigma[1]
        reps.append(rp); rngs.append(rng); outs.append(dict(t=[], p=[], q=[], v=[]))
    for k in range(max(len(st["t"]) for st in streams)):
        ready = []
        for si, (st, rp, rng) in enumerate(zip(streams, reps, rngs)):
            if k >= len(st["t"]):
                continue
            truth = None
            if k <= WINDOW_SIZE:
                dth = rng.normal(size=3) * init_sigma[1]
                truth = (st["gt_p"][k] + rng.normal(size=3) * init_sigma[0], st["gt_R"][k] @ deltaQ_R(dth), st["gt_v"][k] + rng.normal(size=3) * init_sigma[2])
            if rp.begin_frame(st["imu"][k], st["points"][k], st["lines"][k], truth):
                ready.append(si)
        if not ready:
            continue
        # one pre-integration call for every stale IMU buffer of every sequence
        need = [(si, j) for si in ready for j in range(1, WINDOW_SIZE + 1) if reps[si].pre[j] is None]
        if need:
            res = backend.preintegrate([reps[si].bufs[j] for si, j in need])
            for (si, j), r in zip(need, res):
                reps[si].pre[j] = r
        wins = [reps[si].prepare_window() for si in ready]
        results = backend.optimize_many(wins, [reps[si].marg_flag for si in ready], num_iterations, fixed_iterations)
        for si, out in zip(ready, results):
            reps[si].apply_result(out)
            r = reps[si].finish_frame()
            o = outs[si]
            o["t"].append(streams[si]["t"][k]); o["p"].append(r[0]); o["q"].append(r[1]); o["v"].append(r[2])
    return [dict(t=np.array(o["t"]), p=np.array(o["p"]), q=np.array(o["q"]), v=np.array(o["v"]), log=rp.log) for o, rp in zip(outs, reps)]


# ----------------------------------------------------------------------------------------------------------------
# the product back end: C-ABI of include/tcv.h
# ----------------------------------------------------------------------------------------------------------------
class HipBackend:
    """pre-integration, solve, gauge fix and marginalisation on the MI355X through libtcv_hip.so."""

    def __init__(self):
        import tcv
        self.tcv = tcv
        if tcv.lib().tcv_device_count() < 1:
            raise RuntimeError("HipBackend needs a HIP device: the product has no CPU path")

    def preintegrate(self, bufs):
        tcv = self.tcv
        n = len(bufs)
        first, count, rows = [], [], []
        for b in bufs:
            first.append(len(rows)); count.append(len(b["acc"]))
            for a, w in zip(b["acc"], b["gyr"]):
                rows.append([synth.DT_IMU, *a, *w])
        samples = tcv.f64(np.array(rows).reshape(-1, 7))
        init = tcv.f64(np.array([[*b["acc0"], *b["gyr0"], *b["ba"], *b["bg"]] for b in bufs]))
        noise = tcv.f64([synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W])
        fi, ci = tcv.i32(first), tcv.i32(count)
        out = (tcv.ImuPreintegration * n)()
        tcv.check(tcv.lib().tcv_preintegrate(n, tcv.iptr(fi), tcv.iptr(ci), tcv.dptr(samples), samples.shape[0], tcv.dptr(init), tcv.dptr(noise), out))
        return [dict(delta_p=np.array(o.delta_p), delta_q=np.array(o.delta_q), delta_v=np.array(o.delta_v), lin_ba=np.array(o.linearized_ba),
                     lin_bg=np.array(o.linearized_bg), sum_dt=float(o.sum_dt), jacobian=np.array(o.jacobian).reshape(15, 15),
                     covariance=np.array(o.covariance).reshape(15, 15)) for o in out]

    def set_map(self, lines3d, Rbw, Tbw):
        self.map = (np.asarray(lines3d, dtype=float), np.asarray(Rbw, dtype=float), np.asarray(Tbw, dtype=float))

    def match_lines(self, poses, ex, fov, det_frame, det, map3d=None, th=(0.1745, 0.45)):
        """map3d = (lines3d, Rbw, Tbw) of the calling sequence (every sequence of a lock-step replay has its own prior map);
        None: the map of the last set_map call."""
        lines3d, Rbw, Tbw = [np.asarray(a, dtype=float) for a in map3d] if map3d is not None else self.map
        return self.tcv.match_lines(poses, ex, Rbw, Tbw, synth.K_MAT, int(synth.IMG_W), int(synth.IMG_H), WINDOW_SIZE, lines3d, det_frame, det,
                                    th[0], th[1], in_fov=fov)            # angle_th / overlap_th (estimator.cpp:116-119)

    def optimize(self, win, marg_flag, num_iterations, fixed_iterations):
        return self.optimize_many([win], [marg_flag], num_iterations, fixed_iterations)[0]

    def optimize_many(self, wins, marg_flags, num_iterations, fixed_iterations):
        """all windows in device-resident batches: solve -> gauge fix -> marginalisation, results and priors back to the host.
        Windows that marginalise (MARGIN_OLD, or MARGIN_SECOND_NEW with para_Pose[WINDOW_SIZE-1] in the prior) and windows that
        only solve go to separate batches (a batch either carries marginalisation problems for all its windows or for none)."""
        tcv = self.tcv
        n = len(wins)
        Ws = [tcv.Window(w) for w in wins]
        Wn = wins[0]["pose"].shape[0] - 1
        do_marg = []
        for w, flag in zip(wins, marg_flags):
            pb = [tuple(b) for b in w["prior"]["blocks"]] if w.get("prior") is not None else []
            do_marg.append(flag == MARGIN_OLD or ("pose", Wn - 1) in pb)
        outs = [None] * n
        for group in (True, False):
            idx = [i for i in range(n) if do_marg[i] == group]
            if not idx:
                continue
            if group:
                Ms, drops = [], []
                for i in idx:
                    if marg_flags[i] == MARGIN_OLD:
                        mw = tcv.margin_old_window(wins[i]); Ms.append(tcv.Window(mw, share=Ws[i], prior=Ws[i].prior)); drops.append(tcv.margin_old_drops(Ws[i], mw))
                    else:
                        mw = tcv.margin_second_new_window(wins[i]); Ms.append(tcv.Window(mw, share=Ws[i], prior=Ws[i].prior)); drops.append(tcv.margin_second_new_drops(Ws[i]))
                b = tcv.Batch([Ws[i] for i in idx], Ms, drops)
            else:
                b = tcv.Batch([Ws[i] for i in idx])
            b.solve(tcv.default_options(num_iterations, fixed_iterations))
            b.gauge_fix()
            if group:
                b.marginalize()
            b.synchronize(); b.download_states()
            s = b.summaries()
            for k, i in enumerate(idx):
                W = Ws[i]
                out = dict(pose=W.pose.copy(), sb=W.sb.copy(), ex=W.ex.copy(), lam=W.lam.copy(), iterations=s[k].num_iterations, final_cost=s[k].final_cost, prior="keep")
                if group:
                    P = b.prior(k)
                    d = P.export()
                    shift = (lambda nm, j: (nm, j - 1) if nm in ("pose", "sb") else (nm, j)) if marg_flags[i] == MARGIN_OLD else \
                            (lambda nm, j: (nm, j - 1) if (nm in ("pose", "sb") and j == Wn) else (nm, j))
                    d["blocks"] = tcv.prior_blocks(P, W, shift)
                    out["prior"] = d
                outs[i] = out
        return outs


# ----------------------------------------------------------------------------------------------------------------
# the same window management in native code: include/tcv_estimator.h (tc-viml_amd/csrc/tcv_estimator.cpp)
# ----------------------------------------------------------------------------------------------------------------
class _EstimatorConfig(C.Structure):
    _fields_ = [("focal_length", C.c_double), ("min_parallax", C.c_double), ("init_depth", C.c_double),
                ("acc_n", C.c_double), ("gyr_n", C.c_double), ("acc_w", C.c_double), ("gyr_w", C.c_double),
                ("gravity", C.c_double * 3), ("imu_dt", C.c_double), ("K", C.c_double * 9), ("width", C.c_int), ("height", C.c_int),
                ("tic", C.c_double * 3), ("ric", C.c_double * 9), ("estimate_extrinsic", C.c_int),
                ("angle_th", C.c_double), ("overlap_th", C.c_double), ("dist_th", C.c_double),
                ("num_iterations", C.c_int), ("fixed_iterations", C.c_int), ("line_exact_jacobian", C.c_int), ("pad_", C.c_int),
                ("solver_time", C.c_double)]


class _EstimatorStats(C.Structure):
    _fields_ = [("marg_flag", C.c_int), ("n_landmarks", C.c_int), ("n_proj", C.c_int), ("n_line", C.c_int), ("n_line_obs", C.c_int),
                ("iterations", C.c_int), ("prior_n", C.c_int), ("termination", C.c_int), ("final_cost", C.c_double)]


class _FrameInput(C.Structure):
    _fields_ = [("n_imu", C.c_int), ("acc", C.POINTER(C.c_double)), ("gyr", C.POINTER(C.c_double)),
                ("n_points", C.c_int), ("point_ids", C.POINTER(C.c_int)), ("points", C.POINTER(C.c_double)),
                ("n_lines", C.c_int), ("line_ids", C.POINTER(C.c_int)), ("lines", C.POINTER(C.c_double)),
                ("truth", C.POINTER(C.c_double))]


class NativeLockstep:
    """Several sequences advanced in lock step through the native estimator (include/tcv_estimator.h): one estimator per stream,
    every frame the full windows of all streams form ONE device batch (tcv_estimators_optimize).  Python only feeds the per-frame
    streams.  `step(k)` = frame k of every stream: begin_frame for all, one optimize call, finish_frame; returns the number of
    windows optimised.  Same perturbation draws as `run` / `run_many`."""

    def __init__(self, streams, num_iterations: int = 8, fixed_iterations: bool = False, init_sigma=(0.02, 0.005, 0.05), bias_sigma=(0.005, 0.0005),
                 exact_line_jacobian: bool = False, estimate_extrinsic: bool = True, solver_time: float = 0.0):
        """solver_time: SOLVER_TIME of the reference's configuration (sensor.yaml:85; 0: no clock)"""
        import tcv
        self.tcv = tcv
        L = self.L = tcv.lib()
        vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
        self.vp, self.dp, self.ip = vp, dp, ip
        L.tcv_estimator_create.argtypes = [C.POINTER(vp), C.POINTER(_EstimatorConfig)]
        L.tcv_estimator_destroy.argtypes = [vp]; L.tcv_estimator_destroy.restype = None
        L.tcv_estimator_set_biases.argtypes = [vp, dp, dp]
        L.tcv_estimator_set_line_map.argtypes = [vp, C.c_int, dp, dp, dp]
        L.tcv_estimator_begin_frame.argtypes = [vp, C.c_int, dp, dp, C.c_int, ip, dp, C.c_int, ip, dp, dp, ip]
        L.tcv_estimators_begin_frames.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(_FrameInput), ip, ip]
        L.tcv_estimators_optimize.argtypes = [C.POINTER(vp), C.c_int]
        L.tcv_estimators_optimize_begin.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp)]
        L.tcv_estimators_optimize_end.argtypes = [vp]
        L.tcv_thread_stream_slot.argtypes = [C.c_int]
        self.slot = 0                # which of the calling thread's two library streams this object's frames use (tcv_thread_stream_slot)
        self._pending = None
        L.tcv_estimator_finish_frame.argtypes = [vp, dp, dp, dp]
        L.tcv_estimators_finish_frames.argtypes = [C.POINTER(vp), C.c_int, dp, dp, dp, C.POINTER(C.c_int), C.POINTER(_EstimatorStats)]
        L.tcv_estimator_get_stats.argtypes = [vp, C.POINTER(_EstimatorStats)]
        if L.tcv_device_count() < 1:
            raise RuntimeError("the native estimator needs a HIP device: the product has no CPU path")
        cfg = _EstimatorConfig()
        cfg.focal_length = synth.FOCAL_LENGTH; cfg.min_parallax = MIN_PARALLAX; cfg.init_depth = INIT_DEPTH
        cfg.acc_n, cfg.gyr_n, cfg.acc_w, cfg.gyr_w = synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W
        cfg.gravity[:] = list(G); cfg.imu_dt = synth.DT_IMU; cfg.K[:] = list(synth.K_MAT.reshape(9)); cfg.width = int(synth.IMG_W); cfg.height = int(synth.IMG_H)
        cfg.tic[:] = list(synth.TIC); cfg.ric[:] = list(synth.RIC.reshape(9)); cfg.estimate_extrinsic = int(estimate_extrinsic)
        cfg.angle_th, cfg.overlap_th, cfg.dist_th = 0.1745, 0.45, 50.0
        cfg.num_iterations = num_iterations; cfg.fixed_iterations = int(fixed_iterations); cfg.line_exact_jacobian = int(exact_line_jacobian)
        cfg.solver_time = float(solver_time)
        self.streams, self.init_sigma = streams, init_sigma
        self.ests, self.rngs, self.outs = [], [], []
        P = self._P
        for st in streams:
            h = vp()
            cfg.angle_th, cfg.overlap_th, cfg.dist_th = st.get("line_th", (0.1745, 0.45, 50.0))
            tcv.check(L.tcv_estimator_create(C.byref(h), C.byref(cfg)))
            self.ests.append(h)
            rng = np.random.Generator(np.random.PCG64(0xABCD))
            ba = self._f64(st["ba"] + rng.normal(size=3) * bias_sigma[0]); bg = self._f64(st["bg"] + rng.normal(size=3) * bias_sigma[1])
            tcv.check(L.tcv_estimator_set_biases(h, P(ba), P(bg)))
            if "map_lines" in st:
                ml = self._f64(st["map_lines"]); Rb = self._f64(st["Rbw"]).reshape(9); Tb = self._f64(st["Tbw"])
                tcv.check(L.tcv_estimator_set_line_map(h, ml.shape[0], P(ml), P(Rb), P(Tb)))
            self.rngs.append(rng); self.outs.append(dict(t=[], p=[], q=[], v=[], log=[]))
        self.n_frames = max(len(st["t"]) for st in streams)
        self._frames = [dict() for _ in streams]      # per stream: frame -> the begin_frame arguments as C-contiguous arrays (prepare())
        self.host_s = [0.0, 0.0, 0.0, 0]              # seconds in begin_frame x streams | tcv_estimators_optimize | stats + finish_frame x streams (incl. this harness); calls
        self._bias_sigma = bias_sigma
        self._raw = []
        self._batches = {}

    def reset(self):
        """tcv_estimator_reset on every estimator (Estimator::clearState + setParameter) and the replay's own bookkeeping back to frame 0:
        stepping through the streams again must reproduce the first pass bit for bit"""
        L, P = self.L, self._P
        L.tcv_estimator_reset.argtypes = [self.vp]
        self.rngs, self.outs = [], []
        for st, h in zip(self.streams, self.ests):
            self.tcv.check(L.tcv_estimator_reset(h))
            rng = np.random.Generator(np.random.PCG64(0xABCD))
            ba = self._f64(st["ba"] + rng.normal(size=3) * self._bias_sigma[0]); bg = self._f64(st["bg"] + rng.normal(size=3) * self._bias_sigma[1])
            self.tcv.check(L.tcv_estimator_set_biases(h, P(ba), P(bg)))
            self.rngs.append(rng); self.outs.append(dict(t=[], p=[], q=[], v=[], log=[]))
        self._frames = [dict() for _ in self.streams]
        self.host_s = [0.0, 0.0, 0.0, 0]
        self._raw = []
        self._batches = {}

    def prepare(self, k0: int = 0, k1: int = None):
        """converts the per-frame front-end records (dicts / lists of the simulated streams) of frames [k0, k1) into the contiguous arrays
        tcv_estimator_begin_frame takes, ahead of time: what a front end would hand over anyway, kept out of a timed replay loop"""
        for si, st in enumerate(self.streams):
            for k in range(k0, min(len(st["t"]), self.n_frames if k1 is None else k1)):
                if k not in self._frames[si]:
                    self._frames[si][k] = self._frame_args(st, k)

    def _frame_args(self, st, k):
        f64 = self._f64
        imu = st["imu"][k]
        acc = f64(imu[0]) if imu is not None else None; gyr = f64(imu[1]) if imu is not None else None
        pts = st["points"][k]
        ids = np.ascontiguousarray(list(pts.keys()), dtype=np.int32); pv = f64(np.array(list(pts.values())).reshape(-1, 3))
        ln = st["lines"][k]
        if "map_lines" in st:
            lid = np.ascontiguousarray([a for a, _ in ln], dtype=np.int32); lv = f64(np.array([v for _, v in ln]).reshape(-1, 4))
        else:
            lid = np.zeros(len(ln), np.int32); lv = f64(np.array([np.concatenate(t3) for t3 in ln]).reshape(-1, 9))
        # the ctypes arguments of tcv_estimator_begin_frame, made once (the arrays ride along: they own the memory)
        P, ip = self._P, self.ip
        args = (0 if acc is None else acc.shape[0] - 1, None if acc is None else P(acc), None if gyr is None else P(gyr), len(ids), ids.ctypes.data_as(ip), P(pv),
                len(ln), lid.ctypes.data_as(ip), P(lv))
        return args, (acc, gyr, ids, pv, lid, lv)

    @staticmethod
    def _f64(a):
        return np.ascontiguousarray(a, dtype=np.float64)

    def _P(self, a):
        return a.ctypes.data_as(self.dp)

    def _frame_batch(self, k: int):
        """the input records of frame k for every stream that has one: (stream indices, estimator array, tcv_frame_input array, ready array,
        the objects that own the memory)"""
        f64, P, vp = self._f64, self._P, self.vp
        live, keep = [], []
        for si, (st, rng) in enumerate(zip(self.streams, self.rngs)):
            if k >= len(st["t"]):
                continue
            truth = None
            if k <= WINDOW_SIZE:
                dth = rng.normal(size=3) * self.init_sigma[1]
                truth = f64(np.concatenate([st["gt_p"][k] + rng.normal(size=3) * self.init_sigma[0], (st["gt_R"][k] @ deltaQ_R(dth)).reshape(9),
                                            st["gt_v"][k] + rng.normal(size=3) * self.init_sigma[2]]))
            args, own = self._frames[si].pop(k, None) or self._frame_args(st, k)
            live.append((si, args, truth)); keep.append((own, truth))
        n = len(live)
        rec = (_FrameInput * max(1, n))()
        for j, (si, a, truth) in enumerate(live):
            r = rec[j]
            r.n_imu = a[0]
            if a[1] is not None:
                r.acc, r.gyr = a[1], a[2]
            r.n_points, r.point_ids, r.points = a[3], a[4], a[5]
            r.n_lines, r.line_ids, r.lines = a[6], a[7], a[8]
            if truth is not None:
                r.truth = P(truth)
        arr = (vp * max(1, n))(*[self.ests[si] for si, _, _ in live])
        return [si for si, _, _ in live], arr, rec, (C.c_int * max(1, n))(), keep

    def prepare_batches(self, k0: int, k1: int):
        """the tcv_frame_input records of frames [k0, k1) ahead of time (what a front end hands over anyway), kept out of a timed loop.
        Only valid behind the window fill: the initial-pose draws of frames <= WINDOW_SIZE come from the streams' generators in frame order."""
        assert k0 > WINDOW_SIZE
        self.prepare(k0, k1)
        for k in range(k0, k1):
            if k not in self._batches:
                self._batches[k] = self._frame_batch(k)

    def step(self, k: int) -> int:
        self.step_begin(k)
        return self.step_end()

    def step_begin(self, k: int):
        """first half of frame k: begin_frame of every stream and everything of tcv_estimators_optimize up to the last command on the device
        (tcv_estimators_optimize_begin).  A host thread that alternates between two NativeLockstep objects (slot 0 / 1) overlaps the host side
        of one with the kernels of the other."""
        prev = self.L.tcv_thread_stream_slot(self.slot)      # (the slot is per-thread state of the library: put back what the thread had)
        try:
            self._step_begin(k)
        finally:
            self.L.tcv_thread_stream_slot(prev)

    def _step_begin(self, k: int):
        tcv, L, vp = self.tcv, self.L, self.vp
        assert self._pending is None
        t_a = time.perf_counter()
        live, arr_all, rec, rdy, keep = self._batches.pop(k, None) or self._frame_batch(k)
        if not live:
            return
        tcv.check(L.tcv_estimators_begin_frames(arr_all, len(live), rec, rdy, None))
        ready = [si for j, si in enumerate(live) if rdy[j]]
        t_b = time.perf_counter()
        self.host_s[0] += t_b - t_a
        if not ready:
            return
        arr = arr_all if len(ready) == len(live) else (vp * len(ready))(*[self.ests[si] for si in ready])
        ticket = vp()
        tcv.check(L.tcv_estimators_optimize_begin(arr, len(ready), C.byref(ticket)))
        self.host_s[1] += time.perf_counter() - t_b
        self._pending = (k, ready, arr, ticket, keep)

    def step_end(self) -> int:
        """second half: waits for the states, applies them (tcv_estimators_optimize_end), finish_frame of every stream; returns the number of windows"""
        if self._pending is None:
            return 0
        prev = self.L.tcv_thread_stream_slot(self.slot)
        try:
            return self._step_end()
        finally:
            self.L.tcv_thread_stream_slot(prev)

    def _step_end(self) -> int:
        tcv, L, P, vp = self.tcv, self.L, self._P, self.vp
        k, ready, arr, ticket, keep = self._pending
        self._pending = None
        t_b = time.perf_counter()
        tcv.check(L.tcv_estimators_optimize_end(ticket))
        t_c = time.perf_counter()
        self.host_s[1] += t_c - t_b; self.host_s[3] += 1
        nr = len(ready)
        pa, qa, va = np.zeros((nr, 3)), np.zeros((nr, 4)), np.zeros((nr, 3))
        rcs = (C.c_int * nr)()
        sts = (_EstimatorStats * nr)()
        tcv.check(L.tcv_estimators_finish_frames(arr, nr, P(pa), P(qa), P(va), rcs, sts))
        self._raw.append((k, ready, pa, qa, va, sts))      # (turned into the per-stream records by results(): nothing of that in the frame loop)
        self.host_s[2] += time.perf_counter() - t_c
        return len(ready)

    def _flush_raw(self):
        for k, ready, pa, qa, va, sts in self._raw:
            for j, si in enumerate(ready):
                s = sts[j]
                o = self.outs[si]
                o["t"].append(self.streams[si]["t"][k]); o["p"].append(pa[j]); o["q"].append(qa[j]); o["v"].append(va[j])
                o["log"].append(dict(flag=s.marg_flag, n_landmarks=s.n_landmarks, n_proj=s.n_proj, n_line=s.n_line, n_line_obs=s.n_line_obs,
                                     iterations=s.iterations, termination=s.termination, final_cost=s.final_cost, prior_n=s.prior_n))
        self._raw = []

    def results(self):
        self._flush_raw()
        return [dict(t=np.array(o["t"]), p=np.array(o["p"]), q=np.array(o["q"]), v=np.array(o["v"]), log=o["log"]) for o in self.outs]

    def close(self):
        for h in self.ests:
            self.L.tcv_estimator_destroy(h)
        self.ests = []

    def __del__(self):
        if getattr(self, "ests", None):
            self.close()


def run_many_native(streams, num_iterations: int = 8, fixed_iterations: bool = False, init_sigma=(0.02, 0.005, 0.05), bias_sigma=(0.005, 0.0005),
                    exact_line_jacobian: bool = False, estimate_extrinsic: bool = True):
    """`run_many` with the window management in native code (tcv_estimator_*, one estimator per stream, lock-step
    tcv_estimators_optimize): Python only feeds the per-frame streams.  Same perturbation draws as `run` / `run_many`."""
    ls = NativeLockstep(streams, num_iterations, fixed_iterations, init_sigma, bias_sigma, exact_line_jacobian, estimate_extrinsic)
    try:
        for k in range(ls.n_frames):
            ls.step(k)
        return ls.results()
    finally:
        ls.close()
