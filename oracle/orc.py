"""ctypes loader for the C oracle (oracle/liborc.so) -- TEST INFRASTRUCTURE ONLY.
Converts the synthetic-window dicts of tc-viml_amd/synth.py into `orc_window` structs."""
from __future__ import annotations
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
IMU_STRIDE = 287
BLK = {"pose": 0, "sb": 1, "ex": 2}
BLK_INV = {0: "pose", 1: "sb", 2: "ex"}
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class OrcWindow(C.Structure):
    _fields_ = [("n_frames", C.c_int), ("n_landmarks", C.c_int), ("n_imu", C.c_int), ("n_proj", C.c_int),
                ("n_line", C.c_int), ("ex_constant", C.c_int),
                ("pose", _dp), ("speedbias", _dp), ("ex_pose", _dp), ("lam", _dp),
                ("imu_i", _ip), ("imu_j", _ip), ("imu_c", _dp), ("imu_sqrt", _dp),
                ("proj_i", _ip), ("proj_j", _ip), ("proj_l", _ip), ("proj_pts", _dp),
                ("proj_sqrt_info", C.c_double), ("proj_loss_a", C.c_double),
                ("line_f", _ip), ("line_c", _dp),
                ("K", C.c_double * 9), ("Ric", C.c_double * 9), ("Tic", C.c_double * 3), ("line_loss_a", C.c_double),
                ("G", C.c_double * 3),
                ("prior_n", C.c_int), ("prior_nblk", C.c_int),
                ("prior_kind", _ip), ("prior_index", _ip), ("prior_size", _ip), ("prior_idx", _ip),
                ("prior_x0", _dp), ("prior_J0", _dp), ("prior_r0", _dp)]


class OrcSummary(C.Structure):
    _fields_ = [("num_iterations", C.c_int), ("termination", C.c_int),
                ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("cost", C.c_double * 128), ("cost_candidate", C.c_double * 128),
                ("model_cost_change", C.c_double * 128), ("radius", C.c_double * 128), ("mu", C.c_double * 128),
                ("rho", C.c_double * 128), ("step_norm", C.c_double * 128),
                ("step_ok", C.c_int * 128), ("dogleg_case", C.c_int * 128),
                ("first_delta", C.c_double * 2048), ("n_local", C.c_int), ("n_cam", C.c_int)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
        _lib.orc_solve.argtypes = [C.POINTER(OrcWindow), C.c_int, C.c_int, C.POINTER(OrcSummary)]
        _lib.orc_imu_sqrt_info.argtypes = [_dp, _dp]
        _lib.orc_preintegrate.argtypes = [_dp, _dp, C.c_int, C.c_double, _dp, _dp, C.c_double, C.c_double,
                                          C.c_double, C.c_double, _dp, _dp]
        _lib.orc_imu_evaluate.argtypes = [_dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(_dp)]
        _lib.orc_proj_evaluate.argtypes = [_dp, _dp, _dp, C.c_double, _dp, _dp, C.c_double, _dp, C.POINTER(_dp)]
        _lib.orc_line_evaluate.argtypes = [_dp, _dp, _dp, _dp, _dp, _dp, _dp]
        _lib.orc_pose_plus.argtypes = [_dp, _dp, _dp]
        _lib.orc_prior_residual.argtypes = [C.POINTER(OrcWindow), _dp]
        _lib.orc_linearize_dense.argtypes = [C.POINTER(OrcWindow), _dp, _dp, _dp, _ip, _ip]
        _lib.orc_marginalize_old.argtypes = [C.POINTER(OrcWindow), _ip, _ip, _ip, _ip, _ip, _ip, _ip, _dp, _dp, _dp, _dp, _dp]
        _lib.orc_proj_td_evaluate.argtypes = [_dp, _dp, _dp, C.c_double, C.c_double, _dp, _dp, _dp, C.c_double, C.c_double, C.c_double, _dp, C.POINTER(_dp)]
        _lib.orc_gauge_fix.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        _lib.orc_eig_sym.argtypes = [C.c_int, _dp, _dp, _dp]
        _lib.orc_loss_correct.restype = C.c_double
    return _lib


def dptr(a):
    return a.ctypes.data_as(_dp)


def iptr(a):
    return a.ctypes.data_as(_ip)


def pack_imu_constants(imu, k=None):
    """(n_imu, 287): dp3 dq4 dv3 ba3 bg3 sum_dt dp_dba dp_dbg dq_dbg dv_dba dv_dbg cov."""
    n = len(imu["frame_i"])
    out = np.zeros((n, IMU_STRIDE))
    out[:, 0:3] = imu["delta_p"]; out[:, 3:7] = imu["delta_q"]; out[:, 7:10] = imu["delta_v"]
    out[:, 10:13] = imu["lin_ba"]; out[:, 13:16] = imu["lin_bg"]; out[:, 16] = imu["sum_dt"]
    J = imu["jacobian"]
    for b, (r, c) in enumerate([(0, 9), (0, 12), (3, 12), (6, 9), (6, 12)]):
        out[:, 17 + 9 * b:26 + 9 * b] = J[:, r:r + 3, c:c + 3].reshape(n, 9)
    out[:, 62:287] = imu["covariance"].reshape(n, 225)
    return out


class Window:
    """Owns the numpy buffers behind an orc_window."""

    def __init__(self, win: dict, ex_constant=False, imu_sqrt=None):
        self.win = win
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        self.pose = f64(win["pose"]).copy(); self.sb = f64(win["speedbias"]).copy()
        self.ex = f64(win["ex_pose"]).copy(); self.lam = f64(win["lam"]).copy()
        im, pr, ln = win["imu"], win["proj"], win["line"]
        self.imu_i = i32(im["frame_i"]); self.imu_j = i32(im["frame_j"]); self.imu_c = f64(pack_imu_constants(im))
        self.imu_sqrt = None if imu_sqrt is None else f64(imu_sqrt)
        self.proj_i = i32(pr["frame_i"]); self.proj_j = i32(pr["frame_j"]); self.proj_l = i32(pr["landmark"])
        self.proj_pts = f64(np.concatenate([pr["pts_i"], pr["pts_j"]], -1))
        self.line_f = i32(ln["frame"])
        self.line_c = f64(np.concatenate([ln["pts_start"], ln["pts_end"], ln["abc"]], -1)) if len(ln["frame"]) else np.zeros((0, 9))
        w = OrcWindow()
        w.n_frames = self.pose.shape[0]; w.n_landmarks = self.lam.shape[0]
        w.n_imu = len(self.imu_i); w.n_proj = len(self.proj_i); w.n_line = len(self.line_f)
        w.ex_constant = int(ex_constant)
        w.pose = dptr(self.pose); w.speedbias = dptr(self.sb); w.ex_pose = dptr(self.ex); w.lam = dptr(self.lam)
        w.imu_i = iptr(self.imu_i); w.imu_j = iptr(self.imu_j); w.imu_c = dptr(self.imu_c)
        w.imu_sqrt = dptr(self.imu_sqrt) if self.imu_sqrt is not None else None
        w.proj_i = iptr(self.proj_i); w.proj_j = iptr(self.proj_j); w.proj_l = iptr(self.proj_l)
        w.proj_pts = dptr(self.proj_pts)
        w.proj_sqrt_info = float(pr["sqrt_info"]); w.proj_loss_a = float(pr["loss_a"] or 0.0)
        w.line_f = iptr(self.line_f); w.line_c = dptr(self.line_c)
        w.K[:] = list(np.asarray(ln["K"]).reshape(9)); w.Ric[:] = list(np.asarray(ln["Ric"]).reshape(9))
        w.Tic[:] = list(np.asarray(ln["Tic"]).reshape(3)); w.line_loss_a = float(ln["loss_a"] or 0.0)
        w.G[:] = list(np.asarray(win["G"]).reshape(3))
        p = win.get("prior")
        if p is not None:
            self.p_kind = i32([BLK[b[0]] for b in p["blocks"]]); self.p_index = i32([b[1] for b in p["blocks"]])
            self.p_size = i32(p["sizes"]); self.p_idx = i32(p["idx"])
            self.p_x0 = f64(np.concatenate([np.atleast_1d(x) for x in p["x0"]]))
            self.p_J0 = np.asfortranarray(p["J0"], dtype=np.float64); self.p_r0 = f64(p["r0"])
            w.prior_n = int(p["n"]); w.prior_nblk = len(p["blocks"])
            w.prior_kind = iptr(self.p_kind); w.prior_index = iptr(self.p_index)
            w.prior_size = iptr(self.p_size); w.prior_idx = iptr(self.p_idx)
            w.prior_x0 = dptr(self.p_x0); w.prior_J0 = self.p_J0.ctypes.data_as(_dp); w.prior_r0 = dptr(self.p_r0)
        self.c = w

    def states(self):
        return dict(pose=self.pose.copy(), sb=self.sb.copy(), ex=self.ex.copy(), lam=self.lam.copy())

    def solve(self, max_num_iterations=8, fixed_iterations=True):
        s = OrcSummary()
        lib().orc_solve(C.byref(self.c), int(max_num_iterations), int(fixed_iterations), C.byref(s))
        return s

    def linearize_dense(self):
        F, L = self.c.n_frames, self.c.n_landmarks
        nmax = 15 * F + 6 + L
        H = np.zeros((nmax, nmax)); g = np.zeros(nmax); cost = C.c_double(); nl = C.c_int(); nc = C.c_int()
        Hf = np.zeros(nmax * nmax)
        lib().orc_linearize_dense(C.byref(self.c), dptr(Hf), dptr(g), C.byref(cost), C.byref(nl), C.byref(nc))
        n = nl.value
        return Hf[:n * n].reshape(n, n).copy(), g[:n].copy(), cost.value, n, nc.value

    def marginalize_old(self):
        F = self.c.n_frames
        nb_max = 2 * F + 1
        kind = np.zeros(nb_max, np.int32); index = np.zeros(nb_max, np.int32); size = np.zeros(nb_max, np.int32)
        idx = np.zeros(nb_max, np.int32); x0 = np.zeros(16 * F + 7)
        nmax = 15 * F + 6
        J0 = np.zeros(nmax * nmax); r0 = np.zeros(nmax); As = np.zeros(nmax * nmax); bs = np.zeros(nmax)
        m = C.c_int(); n = C.c_int(); nb = C.c_int()
        lib().orc_marginalize_old(C.byref(self.c), C.byref(m), C.byref(n), C.byref(nb), iptr(kind), iptr(index),
                                  iptr(size), iptr(idx), dptr(x0), dptr(J0), dptr(r0), dptr(As), dptr(bs))
        n_, nb_ = n.value, nb.value
        blocks = [(BLK_INV[int(kind[k])], int(index[k])) for k in range(nb_)]
        xs, o = [], 0
        for k in range(nb_):
            xs.append(x0[o:o + size[k]].copy()); o += int(size[k])
        prior = dict(n=n_, m=m.value, blocks=blocks, sizes=[int(s) for s in size[:nb_]], idx=[int(i) for i in idx[:nb_]],
                     x0=xs, J0=J0[:n_ * n_].reshape(n_, n_).T.copy(), r0=r0[:n_].copy())
        dbg = dict(A_schur=As[:n_ * n_].reshape(n_, n_).copy(), b_schur=bs[:n_].copy())
        return prior, dbg


def imu_sqrt_info(cov):
    out = np.zeros(225)
    c = np.ascontiguousarray(cov, dtype=np.float64).reshape(225)
    rc = lib().orc_imu_sqrt_info(dptr(c), dptr(out))
    if rc != 0:
        raise FloatingPointError("covariance inverse not positive definite")
    return out.reshape(15, 15)


def gauge_fix(R0, P0, pose, sb):
    """C restatement of Estimator::double2vector() (estimator.cpp:1537-1581) + the next vector2double() quaternion."""
    pose = np.ascontiguousarray(pose, dtype=np.float64); sb = np.ascontiguousarray(sb, dtype=np.float64)
    R0 = np.ascontiguousarray(R0, dtype=np.float64).reshape(9); P0 = np.ascontiguousarray(P0, dtype=np.float64)
    n = pose.shape[0]
    Rs = np.zeros((n, 3, 3)); Ps = np.zeros((n, 3)); Vs = np.zeros((n, 3)); po = np.zeros((n, 7))
    lib().orc_gauge_fix(n, dptr(R0), dptr(P0), dptr(pose), dptr(sb), dptr(Rs), dptr(Ps), dptr(Vs), dptr(po))
    return Rs, Ps, Vs, po
