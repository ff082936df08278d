"""NumPy restatement of the TC-VIML back-end hot path -- TEST INFRASTRUCTURE ONLY.

**Parity unpinned**: the reference ships no tests / golden vectors for this path and neither
Ceres nor Eigen exist in the authoring container; this file is one of two
independent restatements (the other is oracle/tcv_oracle.c) that must agree with each other.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Every function cites the reference lines it follows (paths relative to
/root/reference/vins_estimator/src/).  Quaternions are stored x,y,z,w (memory order of
Eigen::Map<Quaterniond>, factor/pose_local_parameterization.cpp:6).

The trust-region / dogleg loop restates upstream Ceres 2.x defaults (TrustRegionMinimizer,
DoglegStrategy, SchurComplementSolver) -- *not in /root/reference*, unverified here
(SURVEY.md Appendix C).
"""
from __future__ import annotations

import numpy as np

O_P, O_R, O_V, O_BA, O_BG = 0, 3, 6, 9, 12          # parameters.h:66-73


# --------------------------------------------------------------------------------------
# utility/utility.h:15-68 and the Eigen behaviours listed in SURVEY.md Appendix A
# --------------------------------------------------------------------------------------
def qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz])


def qinv(q):                     # Eigen inverse(): conjugate / squaredNorm
    n2 = float(np.dot(q, q))
    return np.array([-q[0], -q[1], -q[2], q[3]]) / n2


def qrot(q, v):                  # Eigen _transformVector: v + 2w(u x v) + 2u x (u x v)
    u = q[:3]
    uv = np.cross(u, v)
    uv = uv + uv
    return v + q[3] * uv + np.cross(u, uv)


def qnormalized(q):
    return q / np.sqrt(np.dot(q, q))


def q2R(q):                      # Eigen toRotationMatrix (no normalisation)
    x, y, z, w = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def deltaQ(theta):               # utility.h:15-28 (NOT normalised)
    return np.array([theta[0] / 2, theta[1] / 2, theta[2] / 2, 1.0])


def skew(v):                     # utility.h:30-38
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def Qleft33(q):                  # utility.h:50-58, bottomRightCorner<3,3>; positify == identity (:40-48)
    return q[3] * np.eye(3) + skew(q[:3])


def Qright33(p):                 # utility.h:60-68
    return p[3] * np.eye(3) - skew(p[:3])


# --------------------------------------------------------------------------------------
# S2: PoseLocalParameterization::Plus   factor/pose_local_parameterization.cpp:3-19
# --------------------------------------------------------------------------------------
def pose_plus(x, delta):
    out = np.empty(7)
    out[0:3] = x[0:3] + delta[0:3]
    out[3:7] = qnormalized(qmul(x[3:7], deltaQ(delta[3:6])))
    return out


# --------------------------------------------------------------------------------------
# I0: IntegrationBase   factor/integration_base.h:13-158
# --------------------------------------------------------------------------------------
def preintegrate(acc, gyr, dt, lin_ba, lin_bg, acc_n, gyr_n, acc_w, gyr_w):
    """acc/gyr: (S+1,3); sample 0 is (acc_0, gyr_0) of the constructor, 1..S are push_back()s."""
    noise = np.zeros((18, 18))                      # :21-27 (ACC_N/GYR_N used for both k and k+1)
    noise[0:3, 0:3] = acc_n * acc_n * np.eye(3)
    noise[3:6, 3:6] = gyr_n * gyr_n * np.eye(3)
    noise[6:9, 6:9] = acc_n * acc_n * np.eye(3)
    noise[9:12, 9:12] = gyr_n * gyr_n * np.eye(3)
    noise[12:15, 12:15] = acc_w * acc_w * np.eye(3)
    noise[15:18, 15:18] = gyr_w * gyr_w * np.eye(3)
    delta_p = np.zeros(3); delta_v = np.zeros(3); delta_q = np.array([0.0, 0, 0, 1])
    jacobian = np.eye(15); covariance = np.zeros((15, 15)); sum_dt = 0.0
    I3 = np.eye(3)
    dts = np.broadcast_to(np.asarray(dt, dtype=float), (acc.shape[0] - 1,))      # push_back(dt, acc, gyr) carries its own dt (:29-36): a scalar or one per sample
    for k in range(acc.shape[0] - 1):
        dt = float(dts[k])
        a0, g0, a1, g1 = acc[k], gyr[k], acc[k + 1], gyr[k + 1]
        # midPointIntegration :54-128
        un_acc_0 = qrot(delta_q, a0 - lin_ba)
        un_gyr = 0.5 * (g0 + g1) - lin_bg
        rq = qmul(delta_q, np.array([un_gyr[0] * dt / 2, un_gyr[1] * dt / 2, un_gyr[2] * dt / 2, 1.0]))
        un_acc_1 = qrot(rq, a1 - lin_ba)
        un_acc = 0.5 * (un_acc_0 + un_acc_1)
        rp = delta_p + delta_v * dt + 0.5 * un_acc * dt * dt
        rv = delta_v + un_acc * dt
        w_x = 0.5 * (g0 + g1) - lin_bg
        R_w_x = skew(w_x); R_a_0_x = skew(a0 - lin_ba); R_a_1_x = skew(a1 - lin_ba)
        R0 = q2R(delta_q); R1 = q2R(rq)
        F = np.zeros((15, 15))
        F[0:3, 0:3] = I3
        F[0:3, 3:6] = -0.25 * R0 @ R_a_0_x * dt * dt + -0.25 * R1 @ R_a_1_x @ (I3 - R_w_x * dt) * dt * dt
        F[0:3, 6:9] = I3 * dt
        F[0:3, 9:12] = -0.25 * (R0 + R1) * dt * dt
        F[0:3, 12:15] = -0.25 * R1 @ R_a_1_x * dt * dt * -dt
        F[3:6, 3:6] = I3 - R_w_x * dt
        F[3:6, 12:15] = -1.0 * I3 * dt
        F[6:9, 3:6] = -0.5 * R0 @ R_a_0_x * dt + -0.5 * R1 @ R_a_1_x @ (I3 - R_w_x * dt) * dt
        F[6:9, 6:9] = I3
        F[6:9, 9:12] = -0.5 * (R0 + R1) * dt
        F[6:9, 12:15] = -0.5 * R1 @ R_a_1_x * dt * -dt
        F[9:12, 9:12] = I3
        F[12:15, 12:15] = I3
        V = np.zeros((15, 18))
        V[0:3, 0:3] = 0.25 * R0 * dt * dt
        V[0:3, 3:6] = 0.25 * -R1 @ R_a_1_x * dt * dt * 0.5 * dt
        V[0:3, 6:9] = 0.25 * R1 * dt * dt
        V[0:3, 9:12] = V[0:3, 3:6]
        V[3:6, 3:6] = 0.5 * I3 * dt
        V[3:6, 9:12] = 0.5 * I3 * dt
        V[6:9, 0:3] = 0.5 * R0 * dt
        V[6:9, 3:6] = 0.5 * -R1 @ R_a_1_x * dt * 0.5 * dt
        V[6:9, 6:9] = 0.5 * R1 * dt
        V[6:9, 9:12] = V[6:9, 3:6]
        V[9:12, 12:15] = I3 * dt
        V[12:15, 15:18] = I3 * dt
        jacobian = F @ jacobian
        covariance = F @ covariance @ F.T + V @ noise @ V.T
        # propagate :148-156
        delta_p, delta_v = rp, rv
        delta_q = qnormalized(rq)
        sum_dt += dt
    return dict(delta_p=delta_p, delta_q=delta_q, delta_v=delta_v, jacobian=jacobian,
                covariance=covariance, sum_dt=sum_dt)


# --------------------------------------------------------------------------------------
# I1: IMUFactor::Evaluate   factor/imu_factor.h:19-181 + integration_base.h:160-186
# --------------------------------------------------------------------------------------
def imu_sqrt_info(cov):
    """imu_factor.h:64  LLT(cov.inverse()).matrixL().transpose()  (upper triangular)."""
    return np.linalg.cholesky(np.linalg.inv(cov)).T


def imu_evaluate(pose_i, sb_i, pose_j, sb_j, pre, G, sqrt_info=None, want_jac=True):
    Pi, Qi = pose_i[0:3], pose_i[3:7]
    Vi, Bai, Bgi = sb_i[0:3], sb_i[3:6], sb_i[6:9]
    Pj, Qj = pose_j[0:3], pose_j[3:7]
    Vj, Baj, Bgj = sb_j[0:3], sb_j[3:6], sb_j[6:9]
    jac = pre["jacobian"]
    dp_dba = jac[O_P:O_P + 3, O_BA:O_BA + 3]; dp_dbg = jac[O_P:O_P + 3, O_BG:O_BG + 3]
    dq_dbg = jac[O_R:O_R + 3, O_BG:O_BG + 3]
    dv_dba = jac[O_V:O_V + 3, O_BA:O_BA + 3]; dv_dbg = jac[O_V:O_V + 3, O_BG:O_BG + 3]
    sum_dt = pre["sum_dt"]
    delta_q = pre["delta_q"]
    # integration_base.h:173-184
    dba = Bai - pre["lin_ba"]; dbg = Bgi - pre["lin_bg"]
    corrected_delta_q = qmul(delta_q, deltaQ(dq_dbg @ dbg))
    corrected_delta_v = pre["delta_v"] + dv_dba @ dba + dv_dbg @ dbg
    corrected_delta_p = pre["delta_p"] + dp_dba @ dba + dp_dbg @ dbg
    Qi_inv = qinv(Qi)
    r = np.empty(15)
    r[O_P:O_P + 3] = qrot(Qi_inv, 0.5 * G * sum_dt * sum_dt + Pj - Pi - Vi * sum_dt) - corrected_delta_p
    r[O_R:O_R + 3] = 2 * qmul(qinv(corrected_delta_q), qmul(Qi_inv, Qj))[:3]
    r[O_V:O_V + 3] = qrot(Qi_inv, G * sum_dt + Vj - Vi) - corrected_delta_v
    r[O_BA:O_BA + 3] = Baj - Bai
    r[O_BG:O_BG + 3] = Bgj - Bgi
    if sqrt_info is None:
        sqrt_info = imu_sqrt_info(pre["covariance"])
    r = sqrt_info @ r
    if not want_jac:
        return r, None
    Ri_inv = q2R(Qi_inv)
    J0 = np.zeros((15, 7))                                          # imu_factor.h:88-113
    J0[O_P:O_P + 3, O_P:O_P + 3] = -Ri_inv
    J0[O_P:O_P + 3, O_R:O_R + 3] = skew(qrot(Qi_inv, 0.5 * G * sum_dt * sum_dt + Pj - Pi - Vi * sum_dt))
    cdq = qmul(delta_q, deltaQ(dq_dbg @ (Bgi - pre["lin_bg"])))
    J0[O_R:O_R + 3, O_R:O_R + 3] = -(_Qleft4(qmul(qinv(Qj), Qi)) @ _Qright4(cdq))[1:4, 1:4]
    J0[O_V:O_V + 3, O_R:O_R + 3] = skew(qrot(Qi_inv, G * sum_dt + Vj - Vi))
    J0 = sqrt_info @ J0
    J1 = np.zeros((15, 9))                                          # :114-142
    J1[O_P:O_P + 3, 0:3] = -Ri_inv * sum_dt
    J1[O_P:O_P + 3, 3:6] = -dp_dba
    J1[O_P:O_P + 3, 6:9] = -dp_dbg
    J1[O_R:O_R + 3, 6:9] = -Qleft33(qmul(qmul(qinv(Qj), Qi), delta_q)) @ dq_dbg     # :127 (un-corrected delta_q)
    J1[O_V:O_V + 3, 0:3] = -Ri_inv
    J1[O_V:O_V + 3, 3:6] = -dv_dba
    J1[O_V:O_V + 3, 6:9] = -dv_dbg
    J1[O_BA:O_BA + 3, 3:6] = -np.eye(3)
    J1[O_BG:O_BG + 3, 6:9] = -np.eye(3)
    J1 = sqrt_info @ J1
    J2 = np.zeros((15, 7))                                          # :143-161
    J2[O_P:O_P + 3, O_P:O_P + 3] = Ri_inv
    J2[O_R:O_R + 3, O_R:O_R + 3] = Qleft33(qmul(qmul(qinv(cdq), Qi_inv), Qj))
    J2 = sqrt_info @ J2
    J3 = np.zeros((15, 9))                                          # :162-177
    J3[O_V:O_V + 3, 0:3] = Ri_inv
    J3[O_BA:O_BA + 3, 3:6] = np.eye(3)
    J3[O_BG:O_BG + 3, 6:9] = np.eye(3)
    J3 = sqrt_info @ J3
    return r, [J0, J1, J2, J3]


def _Qleft4(q):
    M = np.empty((4, 4))
    M[0, 0] = q[3]; M[0, 1:4] = -q[:3]; M[1:4, 0] = q[:3]; M[1:4, 1:4] = q[3] * np.eye(3) + skew(q[:3])
    return M


def _Qright4(p):
    M = np.empty((4, 4))
    M[0, 0] = p[3]; M[0, 1:4] = -p[:3]; M[1:4, 0] = p[:3]; M[1:4, 1:4] = p[3] * np.eye(3) - skew(p[:3])
    return M


# --------------------------------------------------------------------------------------
# P1: ProjectionFactor::Evaluate   factor/projection_factor.cpp:21-124
# --------------------------------------------------------------------------------------
def proj_evaluate(pose_i, pose_j, ex, lam, pts_i, pts_j, sqrt_info_scalar, want_jac=True):
    Pi, Qi = pose_i[0:3], pose_i[3:7]
    Pj, Qj = pose_j[0:3], pose_j[3:7]
    tic, qic = ex[0:3], ex[3:7]
    inv_dep_i = lam
    pts_camera_i = pts_i / inv_dep_i
    pts_imu_i = qrot(qic, pts_camera_i) + tic
    pts_w = qrot(Qi, pts_imu_i) + Pi
    pts_imu_j = qrot(qinv(Qj), pts_w - Pj)
    pts_camera_j = qrot(qinv(qic), pts_imu_j - tic)
    dep_j = pts_camera_j[2]
    r = sqrt_info_scalar * ((pts_camera_j / dep_j)[:2] - pts_j[:2])
    if not want_jac:
        return r, None
    Ri, Rj, ric = q2R(Qi), q2R(Qj), q2R(qic)
    reduce = np.array([[1. / dep_j, 0, -pts_camera_j[0] / (dep_j * dep_j)],
                       [0, 1. / dep_j, -pts_camera_j[1] / (dep_j * dep_j)]])
    reduce = sqrt_info_scalar * reduce
    J0 = np.zeros((2, 7)); J1 = np.zeros((2, 7)); J2 = np.zeros((2, 7))
    jaco_i = np.hstack([ric.T @ Rj.T, ric.T @ Rj.T @ Ri @ -skew(pts_imu_i)])
    J0[:, :6] = reduce @ jaco_i
    jaco_j = np.hstack([ric.T @ -Rj.T, ric.T @ skew(pts_imu_j)])
    J1[:, :6] = reduce @ jaco_j
    tmp_r = ric.T @ Rj.T @ Ri @ ric
    jaco_ex = np.hstack([ric.T @ (Rj.T @ Ri - np.eye(3)),
                         -tmp_r @ skew(pts_camera_i) + skew(tmp_r @ pts_camera_i)
                         + skew(ric.T @ (Rj.T @ (Ri @ tic + Pi - Pj) - tic))])
    J2[:, :6] = reduce @ jaco_ex
    J3 = (reduce @ ric.T @ Rj.T @ Ri @ ric @ pts_i * -1.0 / (inv_dep_i * inv_dep_i)).reshape(2, 1)
    return r, [J0, J1, J2, J3]


# --------------------------------------------------------------------------------------
# L1: LineProjectionFactor::Evaluate   factor/line_projection_factor.cpp:19-120
# (Jacobian is "as written", NOT the derivative of the residual -- SURVEY.md §8(a) L1)
# --------------------------------------------------------------------------------------
def line_evaluate(pose, pts_start, pts_end, abc, K, b_c_R, b_c_T, want_jac=True, exact=False):
    """exact=False: the reference's Jacobian as written (line_projection_factor.cpp:73-116).  exact=True (test oracle of the opt-in
    extension tcv_problem_set_line_jacobian): the derivative of the same residual w.r.t. PoseLocalParameterization's (dp, dtheta)."""
    T_w = pose[0:3]
    R_w = q2R(qnormalized(pose[3:7]))
    R = b_c_R.T @ R_w.T
    t = -R @ T_w - (b_c_R.T @ b_c_T)
    pcs = R @ pts_start + t
    pce = R @ pts_end + t
    si = K @ pcs; ei = K @ pce
    u_s, v_s = si[0] / si[2], si[1] / si[2]
    u_e, v_e = ei[0] / ei[2], ei[1] / ei[2]
    a, b, c = abc
    d = a * a + b * b
    mus = (b * b * u_s - a * b * v_s - a * c) / d
    mvs = (a * a * v_s - a * b * u_s - b * c) / d
    mue = (b * b * u_e - a * b * v_e - a * c) / d
    mve = (a * a * v_e - a * b * u_e - b * c) / d
    r = np.array([np.sqrt((mus - u_s) ** 2 + (mvs - v_s) ** 2),
                  np.sqrt((mue - u_e) ** 2 + (mve - v_e) ** 2)])
    if not want_jac:
        return r, None
    ep = np.array([[-2 / d * ((mus - u_s) * a * a + a * b * (mvs - v_s)),
                    -2 / d * ((mus - u_s) * a * b + b * b * (mvs - v_s))]])
    ep_ = np.array([[-2 / d * ((mue - u_e) * a * a + a * b * (mve - v_e)),
                     -2 / d * ((mue - u_e) * a * b + b * b * (mve - v_e))]])
    fx, fy = K[0, 0], K[1, 1]
    if exact:
        J = np.zeros((2, 7))
        for e, (pc, P, u, v) in enumerate(((pcs, pts_start, u_s, v_s), (pce, pts_end, u_e, v_e))):
            L = a * u + b * v + c
            duv = np.sign(L) / np.sqrt(d) * np.array([a, b])
            pp = np.array([[fx / pc[2], 0, -fx * pc[0] / (pc[2] * pc[2])], [0, fy / pc[2], -fy * pc[1] / (pc[2] * pc[2])]])
            y = R_w.T @ (np.asarray(P) - T_w)
            J[e, :6] = duv @ pp @ np.hstack([-b_c_R.T @ R_w.T, b_c_R.T @ skew(y)])
        return r, [J]
    pps = np.array([[fx / pcs[2], 0, -fx * pcs[0] / (pcs[2] * pcs[2])],
                    [0, fy / pcs[2], -fy * pcs[1] / (pcs[2] * pcs[2])]])
    ppe = np.array([[fx / pce[2], 0, -fx * pce[0] / (pce[2] * pce[2])],
                    [0, fy / pce[2], -fy * pce[1] / (pce[2] * pce[2])]])
    jaco_s = np.hstack([np.eye(3), skew(pcs)])
    jaco_e = np.hstack([np.eye(3), skew(pce)])
    J = np.zeros((2, 7))
    J[0, :6] = (ep @ pps @ jaco_s)[0]
    J[1, :6] = (ep_ @ ppe @ jaco_e)[0]
    return r, [J]


# --------------------------------------------------------------------------------------
# C1: loss + corrector   marginalization_factor.cpp:37-68 ; CauchyLoss per upstream Ceres
# --------------------------------------------------------------------------------------
def cauchy_rho(s, a):
    b = a * a
    c = 1.0 / b
    ssum = 1.0 + s * c
    inv = 1.0 / ssum
    return np.array([b * np.log(ssum), max(np.finfo(float).tiny, inv), -c * (inv * inv)])


def loss_correct(r, Js, loss_a):
    """returns corrected r, corrected Js, block cost (0.5*rho0)."""
    s = float(np.dot(r, r))
    if loss_a is None or loss_a <= 0:
        return r, Js, 0.5 * s
    rho = cauchy_rho(s, loss_a)
    sqrt_rho1 = np.sqrt(rho[1])
    if s == 0.0 or rho[2] <= 0.0:
        residual_scaling, alpha_sq_norm = sqrt_rho1, 0.0
    else:
        D = 1.0 + 2.0 * s * rho[2] / rho[1]
        alpha = 1.0 - np.sqrt(D)
        residual_scaling = sqrt_rho1 / (1 - alpha)
        alpha_sq_norm = alpha / s
    if Js is not None:
        Js = [sqrt_rho1 * (J - alpha_sq_norm * np.outer(r, r @ J)) for J in Js]
    return r * residual_scaling, Js, 0.5 * rho[0]


# --------------------------------------------------------------------------------------
# M0: MarginalizationFactor::Evaluate   marginalization_factor.cpp:335-384
# --------------------------------------------------------------------------------------
def prior_evaluate(prior, blocks_x, want_jac=True):
    n = prior["n"]
    dx = np.zeros(n)
    for k, x in enumerate(blocks_x):
        size = prior["sizes"][k]; idx = prior["idx"][k]; x0 = prior["x0"][k]
        if size != 7:
            dx[idx:idx + size] = x - x0
        else:
            dx[idx:idx + 3] = x[0:3] - x0[0:3]
            dq = qmul(qinv(x0[3:7]), x[3:7])
            dx[idx + 3:idx + 6] = 2.0 * dq[:3]
            if not (dq[3] >= 0):
                dx[idx + 3:idx + 6] = 2.0 * -dq[:3]
    r = prior["r0"] + prior["J0"] @ dx
    if not want_jac:
        return r, None
    Js = []
    for k in range(len(blocks_x)):
        size = prior["sizes"][k]; idx = prior["idx"][k]
        local = 6 if size == 7 else size
        J = np.zeros((n, size))
        J[:, :local] = prior["J0"][:, idx:idx + local]
        Js.append(J)
    return r, Js


# --------------------------------------------------------------------------------------
# G1: problem structure (estimator.cpp:1679-1886).  Camera-side parameter blocks are ordered
# pose0, sb0, pose1, sb1, ..., ex ; landmarks follow.
# --------------------------------------------------------------------------------------
class Problem:
    def __init__(self, win, ex_constant=False, const_blocks=()):
        self.win = win
        F = win["pose"].shape[0]
        L = win["lam"].shape[0]
        self.F, self.L = F, L
        self.blocks = []                       # (name, idx, gsize)
        for i in range(F):
            self.blocks.append(("pose", i, 7)); self.blocks.append(("sb", i, 9))
        self.blocks.append(("ex", 0, 7))
        self.has_td = win.get("td") is not None            # ESTIMATE_TD: para_Td[0], estimator.cpp:1703-1707
        if self.has_td:
            self.blocks.append(("td", 0, 1))
        # relocalisation (estimator.cpp:1854-1886): `relo_Pose` is one more pose block with a PoseLocalParameterization (:1857-1858), and every
        # landmark that starts at or before the matched frame and has a match in the loop-closure frame gets one more ProjectionFactor on
        # (para_Pose[start], relo_Pose, para_Ex_Pose[0], para_Feature[feature_index]) (:1876-1880).  win["relo"] = dict(pose (7), frame_i, landmark,
        # pts_i, pts_j): the factor list as the reference's loop would build it.
        self.has_relo = win.get("relo") is not None
        if self.has_relo:
            self.blocks.append(("relo", 0, 7))
        for l in range(L):
            self.blocks.append(("lam", l, 1))
        self.const = set(const_blocks)
        if ex_constant:
            self.const.add(("ex", 0))
        self.loff = {}
        off = 0
        for (nm, i, g) in self.blocks:
            if (nm, i) in self.const:
                self.loff[(nm, i)] = -1
                continue
            self.loff[(nm, i)] = off
            off += 6 if g == 7 else g
        self.nlocal = off
        self.nc = off - sum(1 for l in range(L) if ("lam", l) not in self.const)

    def x0(self):
        w = self.win
        x = dict(pose=w["pose"].copy(), sb=w["speedbias"].copy(), ex=w["ex_pose"].copy(), lam=w["lam"].copy())
        if self.has_td:
            x["td"] = np.array([float(w["td"])])
        if self.has_relo:
            x["relo"] = np.asarray(w["relo"]["pose"], dtype=float).copy()
        return x

    @staticmethod
    def get(x, nm, i):
        if nm == "pose": return x["pose"][i]
        if nm == "sb": return x["sb"][i]
        if nm == "ex": return x["ex"]
        if nm == "td": return x["td"]
        if nm == "relo": return x["relo"]
        return x["lam"][i:i + 1]

    def factors(self):
        """list of (kind, index, [(name, idx)...])  in estimator.cpp order: prior, imu, points, lines."""
        w = self.win
        out = []
        if w.get("prior") is not None:
            out.append(("prior", 0, list(w["prior"]["blocks"])))
        im = w["imu"]
        for k in range(len(im["frame_i"])):
            if im["sum_dt"][k] > 10.0:              # estimator.cpp:1726
                continue
            i, j = int(im["frame_i"][k]), int(im["frame_j"][k])
            out.append(("imu", k, [("pose", i), ("sb", i), ("pose", j), ("sb", j)]))
        pr = w["proj"]
        for k in range(len(pr["frame_i"])):
            blks = [("pose", int(pr["frame_i"][k])), ("pose", int(pr["frame_j"][k])), ("ex", 0), ("lam", int(pr["landmark"][k]))]
            out.append(("proj_td", k, blks + [("td", 0)]) if self.has_td else ("proj", k, blks))      # estimator.cpp:1757-1768
        ln = w["line"]
        for k in range(len(ln["frame"])):
            out.append(("line", k, [("pose", int(ln["frame"][k]))]))
        if self.has_relo:                           # estimator.cpp:1854-1886, behind the line factors
            rl = w["relo"]
            for k in range(len(rl["frame_i"])):
                out.append(("proj_relo", k, [("pose", int(rl["frame_i"][k])), ("relo", 0), ("ex", 0), ("lam", int(rl["landmark"][k]))]))
        return out

    def eval_factor(self, fac, x, want_jac=True, imu_sqrt=None):
        kind, k, blks = fac
        w = self.win
        xs = [self.get(x, nm, i) for (nm, i) in blks]
        if kind == "imu":
            im = w["imu"]
            pre = dict(delta_p=im["delta_p"][k], delta_q=im["delta_q"][k], delta_v=im["delta_v"][k],
                       lin_ba=im["lin_ba"][k], lin_bg=im["lin_bg"][k], sum_dt=float(im["sum_dt"][k]),
                       jacobian=im["jacobian"][k], covariance=im["covariance"][k])
            si = None if imu_sqrt is None else imu_sqrt[k]
            r, Js = imu_evaluate(xs[0], xs[1], xs[2], xs[3], pre, w["G"], sqrt_info=si, want_jac=want_jac)
            return loss_correct(r, Js, None)
        if kind == "proj":
            pr = w["proj"]
            r, Js = proj_evaluate(xs[0], xs[1], xs[2], float(xs[3][0]), pr["pts_i"][k], pr["pts_j"][k],
                                  pr["sqrt_info"], want_jac)
            return loss_correct(r, Js, pr["loss_a"])
        if kind == "proj_relo":                     # ProjectionFactor(pts_i, pts_j) with relo_Pose as the second pose, the same CauchyLoss (:1878-1880)
            pr, rl = w["proj"], w["relo"]
            r, Js = proj_evaluate(xs[0], xs[1], xs[2], float(xs[3][0]), rl["pts_i"][k], rl["pts_j"][k], pr["sqrt_info"], want_jac)
            return loss_correct(r, Js, pr["loss_a"])
        if kind == "proj_td":
            pr = w["proj"]
            r, Js = proj_td_evaluate(xs[0], xs[1], xs[2], float(xs[3][0]), float(xs[4][0]), pr["pts_i"][k], pr["pts_j"][k], pr["vel_i"][k], pr["vel_j"][k],
                                     float(pr["td_i"][k]), float(pr["td_j"][k]), float(pr["row_i"][k]), float(pr["row_j"][k]), pr["sqrt_info"],
                                     float(pr["TR"]), float(pr["ROW"]), want_jac)
            return loss_correct(r, Js, pr["loss_a"])
        if kind == "line":
            ln = w["line"]
            r, Js = line_evaluate(xs[0], ln["pts_start"][k], ln["pts_end"][k], ln["abc"][k], ln["K"], ln["Ric"],
                                  ln["Tic"], want_jac, exact=bool(ln.get("exact_jacobian", False)))
            return loss_correct(r, Js, ln["loss_a"])
        if kind == "prior":
            r, Js = prior_evaluate(w["prior"], xs, want_jac)
            return loss_correct(r, Js, None)
        raise ValueError(kind)

    def linearize(self, x, imu_sqrt=None, want_jac=True):
        """dense local Jacobian (rows x nlocal), residual vector, cost."""
        facs = self.factors()
        rows = []
        cost = 0.0
        rs = []
        for fac in facs:
            r, Js, c = self.eval_factor(fac, x, want_jac, imu_sqrt)
            cost += c
            rs.append(r)
            if want_jac:
                Jrow = np.zeros((len(r), self.nlocal))
                for (nm, i), J in zip(fac[2], Js):
                    lo = self.loff[(nm, i)]
                    if lo < 0:
                        continue
                    ls = 6 if J.shape[1] == 7 else J.shape[1]
                    Jrow[:, lo:lo + ls] += J[:, :ls]      # local J = global J * [I;0]  (S2 ComputeJacobian)
                rows.append(Jrow)
        r = np.concatenate(rs)
        J = np.vstack(rows) if want_jac else None
        return J, r, cost

    def plus(self, x, delta):
        out = dict(pose=x["pose"].copy(), sb=x["sb"].copy(), ex=x["ex"].copy(), lam=x["lam"].copy())
        if self.has_td:
            out["td"] = x["td"].copy()
        if self.has_relo:
            out["relo"] = x["relo"].copy()
        for (nm, i, g) in self.blocks:
            lo = self.loff[(nm, i)]
            if lo < 0:
                continue
            if nm == "pose":
                out["pose"][i] = pose_plus(x["pose"][i], delta[lo:lo + 6])
            elif nm == "ex":
                out["ex"] = pose_plus(x["ex"], delta[lo:lo + 6])
            elif nm == "relo":
                out["relo"] = pose_plus(x["relo"], delta[lo:lo + 6])
            elif nm == "sb":
                out["sb"][i] = x["sb"][i] + delta[lo:lo + 9]
            elif nm == "td":
                out["td"] = x["td"] + delta[lo:lo + 1]
            else:
                out["lam"][i] = x["lam"][i] + delta[lo]
        return out

    def ambient(self, x):
        parts = []
        for (nm, i, g) in self.blocks:
            if self.loff[(nm, i)] < 0:
                continue
            parts.append(np.atleast_1d(self.get(x, nm, i)))
        return np.concatenate(parts)


# --------------------------------------------------------------------------------------
# Linear solve: (J'J + mu D^2) y = J'r with landmark Schur elimination + dense Cholesky
# (SPARSE_SCHUR restated densely; any exact elimination order is equivalent)
# --------------------------------------------------------------------------------------
def schur_solve(H, g, nc):
    """H (n x n) SPD-ish with diagonal landmark block H[nc:,nc:]; returns y or None on failure."""
    n = H.shape[0]
    Hcc = H[:nc, :nc]; Hcl = H[:nc, nc:]; hll = np.diag(H)[nc:]
    if np.any(hll <= 0) or not np.all(np.isfinite(hll)):
        return None
    W = Hcl / hll
    S = Hcc - W @ Hcl.T
    rhs = g[:nc] - W @ g[nc:]
    try:
        Lc = np.linalg.cholesky(S)
    except np.linalg.LinAlgError:
        return None
    z = np.linalg.solve(Lc, rhs)
    yc = np.linalg.solve(Lc.T, z)
    yl = (g[nc:] - Hcl.T @ yc) / hll
    y = np.concatenate([yc, yl])
    if not np.all(np.isfinite(y)):
        return None
    return y


# --------------------------------------------------------------------------------------
# Trust-region minimiser with traditional dogleg -- upstream Ceres 2.x defaults, unverified
# here; options set by the reference: estimator.cpp:1888-1897.
# --------------------------------------------------------------------------------------
# Every Ceres default the restatement relies on (Solver::Options / DoglegStrategy constants of upstream Ceres 2.x),
# by name.  solve() uses exactly these unless `ceres_defaults` overrides some: tests/dev/ceres_logic_sensitivity.py perturbs them ONE at a
# time so that a maintainer with a real Ceres can falsify the restatement with one number (iterations to converge / final cost).
CERES_DEFAULTS = dict(
    jacobi_scaling="1/(1+norm)",          # Solver::Options::jacobi_scaling = true: 1 / (1 + ||column||); alternatives "off", "1/norm"
    min_lm_diagonal=1e-6, max_lm_diagonal=1e32,
    initial_trust_region_radius=1e4,
    min_mu=1e-8, max_mu=1.0, mu_increase_factor=10.0,
    mu_decrease="max(min_mu, 2 mu / factor)",   # DoglegStrategy::StepAccepted; alternative "keep"
    decrease_threshold=0.25, increase_threshold=0.75, radius_increase_factor=3.0,
    min_relative_decrease=1e-3,
    function_tolerance=1e-6, parameter_tolerance=1e-8, gradient_tolerance=1e-10,
    min_trust_region_radius=1e-32, max_num_consecutive_invalid_steps=5,
)


def solve(prob: Problem, max_num_iterations=8, fixed_iterations=False, imu_sqrt=None, trace=None, ceres_defaults=None):
    C = dict(CERES_DEFAULTS)
    if ceres_defaults:
        unknown = set(ceres_defaults) - set(C)
        assert not unknown, unknown
        C.update(ceres_defaults)
    x = prob.x0()
    nc = prob.nc
    J, r, cost = prob.linearize(x, imu_sqrt)
    # jacobi scaling, computed once at iteration 0
    cn = np.sqrt((J * J).sum(0))
    if C["jacobi_scaling"] == "1/(1+norm)":
        scale = 1.0 / (1.0 + cn)
    elif C["jacobi_scaling"] == "1/norm":
        scale = 1.0 / np.maximum(cn, 1e-300)
    else:
        scale = np.ones_like(cn)
    J = J * scale
    grad_unscaled = (J / scale).T @ r
    summary = dict(initial_cost=cost, iterations=[dict(it=0, cost=cost, step_ok=True)], termination="NO_CONVERGENCE")
    if not fixed_iterations and np.max(np.abs(grad_unscaled)) <= C["gradient_tolerance"]:
        summary["termination"] = "CONVERGENCE_GRADIENT"; summary["final_cost"] = cost
        return x, summary
    radius, mu, reuse, invalid = C["initial_trust_region_radius"], C["min_mu"], False, 0
    min_mu, max_mu, mu_inc = C["min_mu"], C["max_mu"], C["mu_increase_factor"]
    x_norm = np.linalg.norm(prob.ambient(x))
    it = 0
    diag = grad = gn = None
    alpha = 0.0
    while True:
        if it >= max_num_iterations:
            break
        it += 1
        rec = dict(it=it)
        # ---- DoglegStrategy::ComputeStep
        step_valid_ls = True
        if not reuse:
            reuse = True
            diag = np.sqrt(np.clip((J * J).sum(0), C["min_lm_diagonal"], C["max_lm_diagonal"]))
            grad = (J.T @ r) / diag
            Jg = J @ (grad / diag)
            alpha = float(grad @ grad) / float(Jg @ Jg)
            H = J.T @ J
            g = J.T @ r
            y = None
            while mu < max_mu:
                y = schur_solve(H + np.diag(mu * diag * diag), g, nc)
                if y is None:
                    mu *= mu_inc
                    continue
                break
            if y is None:
                step_valid_ls = False
            else:
                gn = -diag * y
        rec["mu"] = mu
        if step_valid_ls:
            gnorm = np.linalg.norm(grad); gn_norm = np.linalg.norm(gn)
            if gn_norm <= radius:
                step = gn.copy(); step_norm = gn_norm; case = 1
            elif gnorm * alpha >= radius:
                step = -(radius / gnorm) * grad; step_norm = radius; case = 2
            else:
                b_dot_a = -alpha * float(grad @ gn)
                a_sq = (alpha * gnorm) ** 2
                bma_sq = a_sq - 2 * b_dot_a + gn_norm ** 2
                c = b_dot_a - a_sq
                d = np.sqrt(c * c + bma_sq * (radius ** 2 - a_sq))
                beta = (d - c) / bma_sq if c <= 0 else (radius * radius - a_sq) / (d + c)
                step = (-alpha * (1.0 - beta)) * grad + beta * gn
                step_norm = np.linalg.norm(step); case = 3
            step = step / diag
            rec["case"] = case
            Jstep = J @ step
            model_cost_change = -float(Jstep @ (r + Jstep / 2.0))
            step_valid = model_cost_change > 0.0
        else:
            step_valid = False
        if not step_valid:
            invalid += 1
            rec["step_ok"] = False; rec["invalid"] = True; rec["cost"] = cost
            summary["iterations"].append(rec)
            if invalid >= C["max_num_consecutive_invalid_steps"]:
                summary["termination"] = "FAILURE"
                break
            mu *= mu_inc; reuse = False       # DoglegStrategy::StepIsInvalid
            continue
        invalid = 0
        delta = step * scale
        x_c = prob.plus(x, delta)
        _, r_c, cost_c = prob.linearize(x_c, imu_sqrt, want_jac=False)
        rec.update(model_cost_change=model_cost_change, cost_candidate=cost_c, radius=radius,
                   step_norm_dogleg=step_norm, delta=delta.copy())
        dxn = np.linalg.norm(prob.ambient(x) - prob.ambient(x_c))
        if not fixed_iterations and dxn <= C["parameter_tolerance"] * (x_norm + C["parameter_tolerance"]):
            rec["cost"] = cost; summary["iterations"].append(rec)
            summary["termination"] = "CONVERGENCE_PARAMETER"
            break
        cost_change = cost - cost_c
        if not fixed_iterations and abs(cost_change) <= C["function_tolerance"] * cost:
            rec["cost"] = cost; summary["iterations"].append(rec)
            summary["termination"] = "CONVERGENCE_FUNCTION"
            break
        rho = cost_change / model_cost_change
        rec["rho"] = rho
        if rho > C["min_relative_decrease"]:
            x = x_c
            x_norm = np.linalg.norm(prob.ambient(x))
            J, r, cost = prob.linearize(x, imu_sqrt)
            grad_unscaled = J.T @ r
            J = J * scale
            rec["step_ok"] = True
            if rho < C["decrease_threshold"]:
                radius *= 0.5
            if rho > C["increase_threshold"]:
                radius = max(radius, C["radius_increase_factor"] * step_norm)
            if C["mu_decrease"] != "keep":
                mu = max(min_mu, 2.0 * mu / mu_inc)
            reuse = False
            rec["cost"] = cost
            summary["iterations"].append(rec)
            if not fixed_iterations and np.max(np.abs(grad_unscaled)) <= C["gradient_tolerance"]:
                summary["termination"] = "CONVERGENCE_GRADIENT"
                break
        else:
            rec["step_ok"] = False
            radius *= 0.5
            reuse = True
            rec["cost"] = cost
            summary["iterations"].append(rec)
        if radius < C["min_trust_region_radius"]:
            summary["termination"] = "CONVERGENCE_RADIUS"
            break
    summary["final_cost"] = cost
    summary["scale"] = scale
    return x, summary


# --------------------------------------------------------------------------------------
# M1-M4: marginalisation   marginalization_factor.cpp:89-321 ; estimator.cpp:1907-2046 (MARGIN_OLD)
# Deterministic block order: dropped camera blocks, dropped landmarks, then kept blocks, each
# in problem order (the reference's order is unordered_map/address dependent).
# --------------------------------------------------------------------------------------
def marginalize_old(prob: Problem, x, imu_sqrt=None, eps=1e-8):
    w = prob.win
    facs = []
    for fac in prob.factors():
        kind, k, blks = fac
        if kind == "prior":
            drop = [bi for bi, b in enumerate(blks) if b in (("pose", 0), ("sb", 0))]     # estimator.cpp:1918-1924
            facs.append((fac, drop))
        elif kind == "imu" and blks[0] == ("pose", 0):
            if float(w["imu"]["sum_dt"][k]) < 10.0:                                      # :1936
                facs.append((fac, [0, 1]))
        elif kind in ("proj", "proj_td") and blks[0] == ("pose", 0):                      # :1957-1986
            facs.append((fac, [0, 3]))
    # getParameterBlocks with addr_shift pose i -> i-1, sb i -> i-1, ex -> ex (estimator.cpp:2027-2039)
    return _marginalize(prob, x, facs, lambda nm, i: (nm, i - 1) if nm in ("pose", "sb") else (nm, i), imu_sqrt, eps)


def marginalize_second_new(prob: Problem, x, eps=1e-8):
    """MARGIN_SECOND_NEW (estimator.cpp:2047-2113): only the old prior is re-marginalised, dropping para_Pose[WINDOW_SIZE-1];
    addr_shift maps frame WINDOW_SIZE -> WINDOW_SIZE-1 and every other block to itself (:2084-2104)."""
    W = prob.win["pose"].shape[0] - 1                                                     # WINDOW_SIZE
    facs = []
    for fac in prob.factors():
        kind, k, blks = fac
        if kind == "prior" and ("pose", W - 1) in blks:                                   # :2049-2050
            assert ("sb", W - 1) not in blks                                              # ROS_ASSERT :2060
            facs.append((fac, [bi for bi, b in enumerate(blks) if b == ("pose", W - 1)]))
    if not facs:
        return None, None
    return _marginalize(prob, x, facs, lambda nm, i: (nm, i - 1) if (nm in ("pose", "sb") and i == W) else (nm, i), None, eps)


def _marginalize(prob: Problem, x, facs, shift, imu_sqrt=None, eps=1e-8):
    """MarginalizationInfo::addResidualBlockInfo / preMarginalize / marginalize / getParameterBlocks
    (marginalization_factor.cpp:89-321) for the (factor, drop_set) list `facs`."""
    order = {nm_i: n for n, nm_i in enumerate((nm, i) for (nm, i, g) in prob.blocks)}
    gsize = {(nm, i): g for (nm, i, g) in prob.blocks}
    touched, dropped = set(), set()
    for fac, drop in facs:
        for b in fac[2]:
            touched.add(b)
        for d in drop:
            dropped.add(fac[2][d])
    drop_list = sorted(dropped, key=lambda b: order[b])
    keep_list = sorted(touched - dropped, key=lambda b: order[b])
    idx = {}
    pos = 0
    for b in drop_list + keep_list:
        idx[b] = pos
        pos += 6 if gsize[b] == 7 else gsize[b]
        if b == drop_list[-1]:
            m = pos
    n = pos - m
    A = np.zeros((pos, pos)); bvec = np.zeros(pos)
    for fac, drop in facs:
        r, Js, _ = prob.eval_factor(fac, x, True, imu_sqrt)           # ResidualBlockInfo::Evaluate (:3-69)
        blks = fac[2]
        for i in range(len(blks)):
            si = 6 if gsize[blks[i]] == 7 else gsize[blks[i]]
            Ji = Js[i][:, :si]
            for j in range(i, len(blks)):
                sj = 6 if gsize[blks[j]] == 7 else gsize[blks[j]]
                Jj = Js[j][:, :sj]
                ii, jj = idx[blks[i]], idx[blks[j]]
                if i == j:
                    A[ii:ii + si, jj:jj + sj] += Ji.T @ Jj
                else:
                    A[ii:ii + si, jj:jj + sj] += Ji.T @ Jj
                    A[jj:jj + sj, ii:ii + si] = A[ii:ii + si, jj:jj + sj].T
            bvec[idx[blks[i]]:idx[blks[i]] + si] += Ji.T @ r
    A_full = A.copy(); b_full = bvec.copy()
    Amm = 0.5 * (A[:m, :m] + A[:m, :m].T)
    lam_m, V_m = np.linalg.eigh(Amm)
    inv_m = np.where(lam_m > eps, 1.0 / np.where(lam_m > eps, lam_m, 1.0), 0.0)
    Amm_inv = V_m @ np.diag(inv_m) @ V_m.T
    bmm = bvec[:m]; Amr = A[:m, m:]; Arm = A[m:, :m]; Arr = A[m:, m:]; brr = bvec[m:]
    A2 = Arr - Arm @ Amm_inv @ Amr
    b2 = brr - Arm @ Amm_inv @ bmm
    lam2, V2 = np.linalg.eigh(A2)
    S = np.where(lam2 > eps, lam2, 0.0)
    S_inv = np.where(lam2 > eps, 1.0 / np.where(lam2 > eps, lam2, 1.0), 0.0)
    J0 = np.diag(np.sqrt(S)) @ V2.T
    r0 = np.diag(np.sqrt(S_inv)) @ V2.T @ b2
    blocks, sizes, kidx, x0 = [], [], [], []
    for b in keep_list:
        nm, i = b
        blocks.append(shift(nm, i))
        sizes.append(gsize[b]); kidx.append(idx[b] - m)
        x0.append(np.atleast_1d(Problem.get(x, nm, i)).copy())
    prior = dict(n=n, m=m, blocks=blocks, sizes=sizes, idx=kidx, x0=x0, J0=J0, r0=r0)
    dbg = dict(A=A_full, b=b_full, A_schur=A2, b_schur=b2, drop=drop_list, keep=keep_list, m=m, n=n)
    return prior, dbg


# --------------------------------------------------------------------------------------
# G3: Estimator::double2vector() gauge fix   estimator.cpp:1537-1581
# --------------------------------------------------------------------------------------
def R2ypr(R):                    # utility.h:70-85 (degrees)
    n, o, a = R[:, 0], R[:, 1], R[:, 2]
    y = np.arctan2(n[1], n[0])
    p = np.arctan2(-n[2], n[0] * np.cos(y) + n[1] * np.sin(y))
    r = np.arctan2(a[0] * np.sin(y) - a[1] * np.cos(y), -o[0] * np.sin(y) + o[1] * np.cos(y))
    return np.array([y, p, r]) / np.pi * 180.0


def ypr2R(ypr):                  # utility.h:87-112 (degrees)
    y, p, r = np.asarray(ypr, dtype=float) / 180.0 * np.pi
    Rz = np.array([[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1.0]])
    Ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1.0, 0], [-np.sin(p), 0, np.cos(p)]])
    Rx = np.array([[1.0, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
    return Rz @ Ry @ Rx


def R2q(m):                      # Eigen `Quaterniond q{R}` (vector2double, estimator.cpp:1499), returns x y z w
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


def gauge_fix(R0, P0, pose, sb):
    """double2vector: Rs, Ps, Vs (:1565-1581) and the para_Pose the next vector2double() writes (:1494-1503)."""
    pose = np.asarray(pose, dtype=float); sb = np.asarray(sb, dtype=float)
    a = R2ypr(np.asarray(R0, dtype=float))
    R00 = q2R(pose[0, 3:7])
    b = R2ypr(R00)
    rot = ypr2R([a[0] - b[0], 0.0, 0.0])
    if abs(abs(a[1]) - 90.0) < 1.0 or abs(abs(b[1]) - 90.0) < 1.0:      # :1555-1563
        rot = np.asarray(R0, dtype=float) @ R00.T
    n = pose.shape[0]
    Rs = np.zeros((n, 3, 3)); Ps = np.zeros((n, 3)); Vs = np.zeros((n, 3)); po = np.zeros((n, 7))
    for i in range(n):
        Rs[i] = rot @ q2R(qnormalized(pose[i, 3:7]))
        Ps[i] = rot @ (pose[i, :3] - pose[0, :3]) + np.asarray(P0, dtype=float)
        Vs[i] = rot @ sb[i, :3]
        po[i, :3] = Ps[i]; po[i, 3:] = R2q(Rs[i])
    return Rs, Ps, Vs, po


# --------------------------------------------------------------------------------------
# T1: ProjectionTdFactor::Evaluate   factor/projection_td_factor.cpp:34-140  (<2,7,7,7,1,1>)
# --------------------------------------------------------------------------------------
def proj_td_evaluate(pose_i, pose_j, ex, lam, td, pts_i, pts_j, vel_i, vel_j, td_i, td_j, row_i, row_j, sqrt_info_scalar, TR, ROW,
                     want_jac=True):
    vi = np.array([vel_i[0], vel_i[1], 0.0]); vj = np.array([vel_j[0], vel_j[1], 0.0])            # :11-16
    ri = row_i - ROW / 2; rj = row_j - ROW / 2                                                    # :17-18
    pts_i_td = np.asarray(pts_i, dtype=float) - (td - td_i + TR / ROW * ri) * vi                  # :50-51
    pts_j_td = np.asarray(pts_j, dtype=float) - (td - td_j + TR / ROW * rj) * vj
    r, Js = proj_evaluate(pose_i, pose_j, ex, lam, pts_i_td, pts_j_td, sqrt_info_scalar, want_jac)   # :52-130 are ProjectionFactor's
    if not want_jac:
        return r, None
    Pi, Qi = pose_i[:3], pose_i[3:7]; Pj, Qj = pose_j[:3], pose_j[3:7]; tic, qic = ex[:3], ex[3:7]
    pc_i = pts_i_td / lam
    pc_j = qrot(qinv(qic), qrot(qinv(Qj), qrot(Qi, qrot(qic, pc_i) + tic) + Pi - Pj) - tic)
    dep_j = pc_j[2]
    reduce = sqrt_info_scalar * np.array([[1.0 / dep_j, 0, -pc_j[0] / (dep_j * dep_j)], [0, 1.0 / dep_j, -pc_j[1] / (dep_j * dep_j)]])
    Ri, Rj, ric = q2R(Qi), q2R(Qj), q2R(qic)
    J_td = reduce @ ric.T @ Rj.T @ Ri @ ric @ vi / lam * -1.0 + sqrt_info_scalar * vj[:2]        # :131-136
    return r, Js + [J_td.reshape(2, 1)]


# --------------------------------------------------------------------------------------
# N4: 2D-3D line association   estimator.cpp:385-447 (UpdateLinesInFoV), :615-669 (CalEulerDist), :602-613 (CalAngleDist),
#     :671-885 (LineCorrespondenceInFrame); Line2D feature_manager.cpp:4-73.  The reference mixes float and double here
#     (`float xx, yy`, `float min_dist`, `Eigen::Vector3f error`): the float roundings are reproduced.
# --------------------------------------------------------------------------------------
class Line2D:
    def __init__(self, v):                       # Line2D(const Eigen::Vector4d &Vec), feature_manager.cpp:4-16
        v = np.asarray(v, dtype=float)
        self.PtrStart = v[0:2].copy(); self.PtrEnd = v[2:4].copy()
        self.LineVec = self.PtrEnd - self.PtrStart
        self.Length = np.sqrt(self.LineVec[0] * self.LineVec[0] + self.LineVec[1] * self.LineVec[1])
        self.Direction = self.LineVec / self.Length
        self.A = self.PtrEnd[1] - self.PtrStart[1]
        self.B = self.PtrStart[0] - self.PtrEnd[0]
        self.C = self.PtrEnd[0] * self.PtrStart[1] - self.PtrStart[0] * self.PtrEnd[1]
        self.A2B2 = np.sqrt(self.A * self.A + self.B * self.B)

    def point2flined(self, P):                   # Line2D::Point2Flined, feature_manager.cpp:48-73
        ts = P - self.PtrStart; d1 = np.sqrt(ts[0] * ts[0] + ts[1] * ts[1])
        te = P - self.PtrEnd; d2 = np.sqrt(te[0] * te[0] + te[1] * te[1])
        A_, B_ = self.B, -self.A
        C_ = -1 * (A_ * P[0] + B_ * P[1])
        det = self.A * B_ - self.B * A_          # Cof.inverse() * (-C, -C_): 2x2 adjugate * (1 / det)
        invdet = 1.0 / det
        ix = (B_ * invdet) * (-self.C) + (-self.B * invdet) * (-C_)
        iy = (-A_ * invdet) * (-self.C) + (self.A * invdet) * (-C_)
        if (ix - self.PtrStart[0]) * (ix - self.PtrEnd[0]) >= 0:
            return self.PtrStart.copy() if d1 < d2 else self.PtrEnd.copy()
        return np.array([ix, iy])


def cal_angle_dist(proj: Line2D, det: Line2D):   # estimator.cpp:602-613
    with np.errstate(invalid="ignore"):
        beta = np.arccos(abs(det.Direction[0] * proj.Direction[0] + det.Direction[1] * proj.Direction[1]))
    return np.pi if np.isnan(beta) else beta


def cal_euler_dist(proj: Line2D, det: Line2D):   # estimator.cpp:615-669
    sampleNum = 10
    line1, line2 = (det, proj) if det.Length <= proj.Length else (proj, det)
    d = line2.point2flined(line1.PtrStart) - line2.point2flined(line1.PtrEnd)
    overlap = np.sqrt(d[0] * d[0] + d[1] * d[1]) / line2.Length
    px, py = line1.PtrStart
    step_x = (line1.PtrStart[0] - line1.PtrEnd[0]) / sampleNum; step_y = (line1.PtrStart[1] - line1.PtrEnd[1]) / sampleNum
    dist = 0.0
    for i in range(sampleNum):
        x = px + i * step_x; y = py + i * step_y
        dist = dist + abs(line2.A * x + line2.B * y + line2.C) / line2.A2B2
    dist = dist + 1 * abs(line2.A * line1.PtrStart[0] + line2.B * line1.PtrStart[1] + line2.C) / line2.A2B2
    dist = dist + 1 * abs(line2.A * line1.PtrEnd[0] + line2.B * line1.PtrEnd[1] + line2.C) / line2.A2B2
    dist = dist / (sampleNum + 2)
    if np.isnan(dist) or np.isnan(overlap):
        return 10000.0, 0.0
    return dist, overlap


def _line_extrinsic(pose, ex, Rbw, Tbw):         # estimator.cpp:388-402 / :679-692
    Ric = q2R(qnormalized(ex[3:7])); Tic = ex[:3]
    Rbi = q2R(qnormalized(pose[3:7])); Tbi = pose[:3]
    R = Ric.T @ Rbi.T @ Rbw
    T = Ric.T @ (Rbi.T @ (Tbw - Tbi) - Tic)
    return R, T


def lines_in_fov(pose, ex, Rbw, Tbw, K, width, height, window_size, lines3d):
    """UpdateLinesInFoV (estimator.cpp:385-447): mask of the map lines kept for this frame."""
    R, T = _line_extrinsic(np.asarray(pose, float), np.asarray(ex, float), Rbw, Tbw)
    hu, hd, wl, wr = -2 * window_size, 2 * window_size + height, -2 * window_size, 2 * window_size + width
    out = np.zeros(len(lines3d), dtype=bool)
    for j, l in enumerate(np.asarray(lines3d, float)):
        ps = R @ l[0:3] + T; pe = R @ l[3:6] + T
        if ps[2] > 0 and pe[2] > 0:
            xx = K[0, 0] * ps[0] / ps[2] + K[0, 2]; yy = K[1, 1] * ps[1] / ps[2] + K[1, 2]
            xx_ = K[0, 0] * pe[0] / pe[2] + K[0, 2]; yy_ = K[1, 1] * pe[1] / pe[2] + K[1, 2]
            s = xx > wl and xx < (wr - 1) and yy > hu and yy < hd
            e = xx_ > wl and xx_ < (wr - 1) and yy_ > hu and yy_ < hd
            out[j] = s or e
    return out


def line_correspondence_in_frame(pose, ex, Rbw, Tbw, K, width, height, lines3d, in_fov, det_vec4, angle_th, overlap_th):
    """LineCorrespondenceInFrame (estimator.cpp:671-885).  Returns (error float32[3] = angle, distance, overlap or -1s,
    index into lines3d or -1, projected line as pixel end points)."""
    f32 = np.float32
    R, T = _line_extrinsic(np.asarray(pose, float), np.asarray(ex, float), Rbw, Tbw)
    det = Line2D(det_vec4)
    idxs = np.nonzero(in_fov)[0]
    err = np.array([-1, -1, -1], dtype=np.float32)
    if len(idxs) == 0:
        return err, -1, np.asarray(det_vec4, float).copy()
    min_dist = f32(10000.0); choose = -1; proj_vec = np.asarray(det_vec4, float).copy()
    for j in idxs:
        l = np.asarray(lines3d[j], float)
        ps = R @ l[0:3] + T; pe = R @ l[3:6] + T
        sflag = eflag = False
        xx = yy = xx_ = yy_ = f32(0)
        if ps[2] > 0 and pe[2] > 0:
            xx = f32(K[0, 0] * ps[0] / ps[2] + K[0, 2]); yy = f32(K[1, 1] * ps[1] / ps[2] + K[1, 2])
            xx_ = f32(K[0, 0] * pe[0] / pe[2] + K[0, 2]); yy_ = f32(K[1, 1] * pe[1] / pe[2] + K[1, 2])
            sflag = bool(xx > 0 and xx < width - 1 and yy > 0 and yy < height - 1)
            eflag = bool(xx_ > 0 and xx_ < width - 1 and yy_ > 0 and yy_ < height - 1)
        cand = None
        if sflag and eflag:
            cand = [float(xx), float(yy), float(xx_), float(yy_)]
        elif sflag or eflag:                     # walk the hidden end point back towards the visible one, t = 0.9, 0.8, ...
            a, b = (ps, pe) if sflag else (pe, ps)
            dirvec = b - a
            t = 0.9
            found = False
            while t > 0:
                q = a + t * dirvec
                if q[2] > 0:
                    x = K[0, 0] * q[0] / q[2] + K[0, 2]; y = K[1, 1] * q[1] / q[2] + K[1, 2]
                    if x > 0 and x < (width - 1) and y > 0 and y < (height - 1):
                        found = True
                        break
                t = t - 0.1
            if found:
                cand = [float(xx), float(yy), x, y] if sflag else [x, y, float(xx_), float(yy_)]
        if cand is None:
            continue
        tl = Line2D(cand)
        angle = cal_angle_dist(tl, det)
        if angle > angle_th:
            continue
        dist, ov = cal_euler_dist(tl, det)
        distance = f32(dist); overlap = f32(ov)
        if float(overlap) < overlap_th:
            continue
        if distance < min_dist:
            min_dist = distance; choose = int(j); proj_vec = np.array(cand)
            err = np.array([f32(angle), min_dist, overlap], dtype=np.float32)
    if choose == -1:
        return np.array([-1, -1, -1], dtype=np.float32), -1, np.asarray(det_vec4, float).copy()
    return err, choose, proj_vec
