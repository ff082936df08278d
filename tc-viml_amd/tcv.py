"""ctypes binding of the C-ABI in include/tcv.h (libtcv_hip.so) plus helpers that turn the
synthetic-window dicts of synth.py into `tcv_window_desc` structs.

This is host plumbing only: every compute entry point runs HIP kernels on the GPU and returns
TCV_ERR_NO_DEVICE (-2) when no device is visible -- there is no CPU fallback, and nothing here
imports `oracle/`.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TCV_LIB", os.path.join(_HERE, "libtcv_hip.so"))

TCV_OK, TCV_ERR_INVALID, TCV_ERR_NO_DEVICE, TCV_ERR_TOO_LARGE, TCV_ERR_HIP, TCV_ERR_UNSUPPORTED, TCV_ERR_NUMERIC = 0, -1, -2, -3, -4, -5, -6
TCV_PARAM_EUCLIDEAN, TCV_PARAM_POSE = 0, 1
TCV_MAX_TRACE = 64
TERMINATION = ["NO_CONVERGENCE", "CONVERGENCE_GRADIENT", "CONVERGENCE_PARAMETER", "CONVERGENCE_FUNCTION",
               "CONVERGENCE_RADIUS", "FAILURE"]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

# every symbol include/tcv.h declares (checked by tests/test_abi.py against the header text)
EXPORTS = [
    "tcv_version", "tcv_last_error", "tcv_device_count", "tcv_set_device", "tcv_device_memory_stats",
    "tcv_problem_create", "tcv_problem_destroy", "tcv_problem_add_parameter_block",
    "tcv_problem_set_parameter_block_constant", "tcv_problem_set_gravity", "tcv_problem_add_imu_factor",
    "tcv_problem_add_projection_factor", "tcv_problem_add_projection_td_factor", "tcv_problem_set_rolling_shutter", "tcv_problem_set_line_jacobian", "tcv_problem_add_line_factor", "tcv_problem_add_marginalization_factor",
    "tcv_problem_from_window", "tcv_problem_num_parameter_blocks", "tcv_problem_num_residual_blocks",
    "tcv_problem_num_residuals", "tcv_problem_plan_stats", "tcv_solver_options_default", "tcv_solve", "tcv_marginalize",
    "tcv_prior_create", "tcv_prior_dims", "tcv_prior_export", "tcv_prior_export_schur", "tcv_prior_keep_block_addresses", "tcv_prior_destroy",
    "tcv_batch_create", "tcv_batch_attach_marginalization", "tcv_batch_destroy", "tcv_batch_solve", "tcv_batch_marginalize", "tcv_batch_synchronize",
    "tcv_batch_download_states", "tcv_batch_get_summaries", "tcv_batch_get_prior", "tcv_batch_get_first_step",
    "tcv_batch_plan_stats", "tcv_batch_layout", "tcv_batch_stats", "tcv_batch_size",
    "tcv_eval_imu_factors", "tcv_eval_projection_factors", "tcv_eval_line_factors", "tcv_pose_plus", "tcv_preintegrate",
    "tcv_problem_set_frames", "tcv_gauge_fix", "tcv_batch_gauge_fix", "tcv_batch_set_fused_gauge_fix", "tcv_set_solver_variant", "tcv_set_cooperative", "tcv_batch_cooperative", "tcv_batch_get_priors", "tcv_batch_marg_status", "tcv_eval_projection_td_factors", "tcv_match_lines",
    "tcv_batch_download_priors", "tcv_batch_download_priors_compact",
    "tcv_batch_get_priors_device", "tcv_batch_get_priors_device_async", "tcv_prior_is_device_resident", "tcv_problem_set_marginalization_prior",
    "tcv_problems_set_marginalization_prior", "tcv_priors_destroy",
    "tcv_match_lines_batch", "tcv_preintegrate_device", "tcv_preint_sum_dt", "tcv_preint_export", "tcv_preint_destroy", "tcv_problem_add_imu_factor_device",
    "tcv_microbench_fp64", "tcv_problem_plan_ints", "tcv_set_packer_reference", "tcv_plan_cache_stats", "tcv_problems_pack_bench", "tcv_line_map_create", "tcv_line_map_destroy", "tcv_batch_download_states_brief", "tcv_thread_stream_slot", "tcv_batch_download_states_begin", "tcv_batch_download_states_end",
]


STREAM_THREAD = C.c_void_p(2 ** (8 * C.sizeof(C.c_void_p)) - 1)      # TCV_STREAM_THREAD: the calling thread's own stream


class ImuPreintegration(C.Structure):
    _fields_ = [("delta_p", C.c_double * 3), ("delta_q", C.c_double * 4), ("delta_v", C.c_double * 3),
                ("linearized_ba", C.c_double * 3), ("linearized_bg", C.c_double * 3), ("sum_dt", C.c_double),
                ("jacobian", C.c_double * 225), ("covariance", C.c_double * 225)]


class SolverOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int), ("max_solver_time_in_seconds", C.c_double),
                ("fixed_iterations", C.c_int), ("workgroups_per_window", C.c_int), ("use_mfma", C.c_int),
                ("threads_per_window", C.c_int), ("record_first_step", C.c_int)]


class SolverSummary(C.Structure):
    _fields_ = [("num_iterations", C.c_int), ("termination", C.c_int),
                ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("cost", C.c_double * TCV_MAX_TRACE), ("cost_candidate", C.c_double * TCV_MAX_TRACE),
                ("model_cost_change", C.c_double * TCV_MAX_TRACE), ("radius", C.c_double * TCV_MAX_TRACE),
                ("mu", C.c_double * TCV_MAX_TRACE), ("rho", C.c_double * TCV_MAX_TRACE),
                ("step_norm", C.c_double * TCV_MAX_TRACE),
                ("step_ok", C.c_int * TCV_MAX_TRACE), ("dogleg_case", C.c_int * TCV_MAX_TRACE)]


class WindowDesc(C.Structure):
    _fields_ = [("n_frames", C.c_int), ("n_landmarks", C.c_int), ("n_imu", C.c_int), ("n_proj", C.c_int),
                ("n_line", C.c_int), ("estimate_extrinsic", C.c_int),
                ("para_pose", _dp), ("para_speedbias", _dp), ("para_ex_pose", _dp), ("para_feature", _dp),
                ("imu_frame_i", _ip), ("imu_frame_j", _ip), ("imu", C.POINTER(ImuPreintegration)),
                ("proj_frame_i", _ip), ("proj_frame_j", _ip), ("proj_feature", _ip), ("proj_pts", _dp),
                ("proj_sqrt_info", C.c_double), ("proj_loss_a", C.c_double),
                ("line_frame", _ip), ("line_data", _dp),
                ("line_K", C.c_double * 9), ("line_Ric", C.c_double * 9), ("line_Tic", C.c_double * 3),
                ("line_loss_a", C.c_double), ("gravity", C.c_double * 3),
                ("prior", C.c_void_p), ("prior_block_kind", _ip), ("prior_block_index", _ip),
                ("para_td", _dp), ("proj_td_aux", _dp), ("td_TR", C.c_double), ("td_ROW", C.c_double),
                ("line_exact_jacobian", C.c_int), ("pad_", C.c_int),
                ("imu_device", C.POINTER(C.c_void_p))]


_lib = None


class TcvError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"tcv status {status}: {msg}")
        self.status = status


def lib():
    """Loads libtcv_hip.so; raises if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} missing: run `python tc-viml_amd/build.py` (HIP extension is mandatory)")
        L = C.CDLL(LIB_PATH)
        L.tcv_version.restype = C.c_char_p
        L.tcv_last_error.restype = C.c_char_p
        vp = C.c_void_p
        L.tcv_problem_create.argtypes = [C.POINTER(vp)]
        L.tcv_problem_destroy.argtypes = [vp]
        L.tcv_problem_destroy.restype = None
        L.tcv_problem_add_parameter_block.argtypes = [vp, _dp, C.c_int, C.c_int]
        L.tcv_problem_set_parameter_block_constant.argtypes = [vp, _dp]
        L.tcv_problem_set_gravity.argtypes = [vp, _dp]
        L.tcv_problem_add_imu_factor.argtypes = [vp, C.POINTER(ImuPreintegration), _dp, _dp, _dp, _dp]
        L.tcv_problem_add_projection_factor.argtypes = [vp, _dp, _dp, C.c_double, C.c_double, _dp, _dp, _dp, _dp]
        L.tcv_problem_add_line_factor.argtypes = [vp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_double, _dp]
        L.tcv_problem_add_projection_td_factor.argtypes = [vp, _dp, _dp, _dp, _dp] + [C.c_double] * 6 + [_dp] * 5
        L.tcv_problem_set_rolling_shutter.argtypes = [vp, C.c_double, C.c_double]
        L.tcv_problem_set_line_jacobian.argtypes = [vp, C.c_int]
        L.tcv_problem_add_marginalization_factor.argtypes = [vp, vp, C.POINTER(_dp), C.c_int]
        L.tcv_problem_from_window.argtypes = [C.POINTER(WindowDesc), C.POINTER(vp)]
        for f in ("tcv_problem_num_parameter_blocks", "tcv_problem_num_residual_blocks", "tcv_problem_num_residuals"):
            getattr(L, f).argtypes = [vp]
        L.tcv_problem_plan_stats.argtypes = [vp, _ip]
        L.tcv_solver_options_default.argtypes = [C.POINTER(SolverOptions)]
        L.tcv_solver_options_default.restype = None
        L.tcv_solve.argtypes = [C.POINTER(SolverOptions), vp, C.POINTER(SolverSummary)]
        L.tcv_marginalize.argtypes = [vp, C.POINTER(_dp), C.c_int, C.POINTER(vp)]
        L.tcv_prior_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, _ip, _ip, _dp, _dp, _dp]
        L.tcv_prior_dims.argtypes = [vp, _ip, _ip, _ip, _ip]
        L.tcv_prior_export.argtypes = [vp, _ip, _ip, _dp, _dp, _dp]
        L.tcv_prior_export_schur.argtypes = [vp, _dp, _dp]
        L.tcv_prior_keep_block_addresses.argtypes = [vp, C.POINTER(_dp)]
        L.tcv_prior_destroy.argtypes = [vp]
        L.tcv_prior_destroy.restype = None
        L.tcv_batch_create.argtypes = [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.POINTER(_dp)), _ip, C.c_int]
        L.tcv_batch_attach_marginalization.argtypes = [vp, C.POINTER(vp), C.POINTER(C.POINTER(_dp)), _ip]
        L.tcv_batch_destroy.argtypes = [vp]
        L.tcv_batch_destroy.restype = None
        L.tcv_batch_solve.argtypes = [vp, C.POINTER(SolverOptions), vp]
        L.tcv_batch_marginalize.argtypes = [vp, vp]
        L.tcv_batch_synchronize.argtypes = [vp]
        L.tcv_batch_download_states.argtypes = [vp]
        L.tcv_batch_get_summaries.argtypes = [vp, C.POINTER(SolverSummary), C.c_int]
        L.tcv_batch_get_prior.argtypes = [vp, C.c_int, C.POINTER(vp)]
        L.tcv_batch_download_priors.argtypes = [vp]
        L.tcv_batch_download_priors_compact.argtypes = [vp]
        L.tcv_batch_get_priors_device.argtypes = [vp, C.POINTER(vp), C.c_int]
        L.tcv_batch_get_priors_device_async.argtypes = [vp, C.POINTER(vp), C.c_int]
        L.tcv_prior_is_device_resident.argtypes = [vp]
        L.tcv_problem_set_marginalization_prior.argtypes = [vp, vp]
        L.tcv_problems_set_marginalization_prior.argtypes = [C.POINTER(vp), C.POINTER(vp), C.c_int]
        L.tcv_priors_destroy.argtypes = [C.POINTER(vp), C.c_int]
        L.tcv_priors_destroy.restype = None
        L.tcv_batch_get_first_step.argtypes = [vp, C.c_int, _dp, C.c_int, _ip]
        L.tcv_batch_plan_stats.argtypes = [vp, _ip, _dp, _ip, _ip]
        L.tcv_batch_layout.argtypes = [vp]
        L.tcv_batch_stats.argtypes = [vp, _dp, _dp, _dp]
        L.tcv_batch_size.argtypes = [vp]
        L.tcv_eval_imu_factors.argtypes = [C.c_int, C.POINTER(ImuPreintegration), _dp, _dp, C.c_int, _dp, _dp, _dp]
        L.tcv_eval_projection_factors.argtypes = [C.c_int, _dp, _dp, C.c_double, _dp, _dp]
        L.tcv_eval_line_factors.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.tcv_pose_plus.argtypes = [C.c_int, _dp, _dp, _dp]
        L.tcv_preintegrate.argtypes = [C.c_int, _ip, _ip, _dp, C.c_int, _dp, _dp, C.POINTER(ImuPreintegration)]
        L.tcv_preintegrate_device.argtypes = [C.c_int, _ip, _ip, _dp, C.c_int, _dp, _dp, C.POINTER(vp)]
        L.tcv_preint_sum_dt.argtypes = [vp]
        L.tcv_preint_sum_dt.restype = C.c_double
        L.tcv_preint_export.argtypes = [vp, C.POINTER(ImuPreintegration)]
        L.tcv_preint_destroy.argtypes = [vp]
        L.tcv_preint_destroy.restype = None
        L.tcv_problem_add_imu_factor_device.argtypes = [vp, vp, _dp, _dp, _dp, _dp]
        L.tcv_problem_set_frames.argtypes = [vp, C.c_int, C.POINTER(_dp), C.POINTER(_dp)]
        L.tcv_gauge_fix.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.tcv_batch_gauge_fix.argtypes = [vp, vp]
        L.tcv_batch_set_fused_gauge_fix.argtypes = [vp, C.c_int]
        L.tcv_batch_marg_status.argtypes = [vp, _ip, C.c_int]
        L.tcv_match_lines.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_int, _ip, _dp, C.c_double,
                                      C.c_double, C.c_int, C.POINTER(C.c_ubyte), _ip, C.POINTER(C.c_float), _dp]
        L.tcv_eval_projection_td_factors.argtypes = [C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, C.c_double, _dp, _dp]
        _lib = L
    return _lib


def check(rc):
    if rc != TCV_OK:
        raise TcvError(rc, lib().tcv_last_error().decode())


def dptr(a):
    return a.ctypes.data_as(_dp)


def iptr(a):
    return a.ctypes.data_as(_ip)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def device_memory_stats():
    """(live_bytes, cached_bytes, live_buffers) of the library's device allocator (tcv_device_memory_stats)"""
    lb, cb, n = C.c_ulonglong(0), C.c_ulonglong(0), C.c_int(0)
    check(lib().tcv_device_memory_stats(C.byref(lb), C.byref(cb), C.byref(n)))
    return int(lb.value), int(cb.value), int(n.value)


def default_options(max_num_iterations=8, fixed_iterations=True, use_mfma=True, threads=256, record_first_step=False, workgroups_per_window=0):
    o = SolverOptions()
    lib().tcv_solver_options_default(C.byref(o))
    o.workgroups_per_window = workgroups_per_window
    o.max_num_iterations = max_num_iterations
    o.fixed_iterations = int(fixed_iterations)
    o.use_mfma = int(use_mfma)
    o.threads_per_window = threads
    o.record_first_step = int(record_first_step)
    return o


def pack_imu(imu) -> C.Array:
    """synth 'imu' dict (arrays over n_imu) -> array of tcv_imu_preintegration."""
    n = len(imu["frame_i"])
    arr = (ImuPreintegration * n)()
    for k in range(n):
        p = arr[k]
        p.delta_p[:] = list(imu["delta_p"][k]); p.delta_q[:] = list(imu["delta_q"][k]); p.delta_v[:] = list(imu["delta_v"][k])
        p.linearized_ba[:] = list(imu["lin_ba"][k]); p.linearized_bg[:] = list(imu["lin_bg"][k])
        p.sum_dt = float(imu["sum_dt"][k])
        p.jacobian[:] = list(np.asarray(imu["jacobian"][k]).reshape(225))
        p.covariance[:] = list(np.asarray(imu["covariance"][k]).reshape(225))
    return arr


class Prior:
    """Owns a tcv_prior handle (MarginalizationInfo layout)."""

    def __init__(self, handle=None):
        self.h = handle

    @classmethod
    def from_dict(cls, p):
        sizes = i32(p["sizes"]); idx = i32(np.asarray(p["idx"]) + int(p.get("m", 0)))   # reference convention: idx includes m
        x0 = f64(np.concatenate([np.atleast_1d(x) for x in p["x0"]]))
        J0 = np.asfortranarray(p["J0"], dtype=np.float64)     # column-major like Eigen::MatrixXd
        r0 = f64(p["r0"])
        h = C.c_void_p()
        check(lib().tcv_prior_create(C.byref(h), int(p.get("m", 0)), int(p["n"]), len(sizes), iptr(sizes), iptr(idx), dptr(x0),
                                     J0.ctypes.data_as(_dp), dptr(r0)))
        return cls(h)

    def on_device(self):
        return bool(lib().tcv_prior_is_device_resident(self.h))

    def dims(self):
        m, n, nb, xs = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(lib().tcv_prior_dims(self.h, C.byref(m), C.byref(n), C.byref(nb), C.byref(xs)))
        return m.value, n.value, nb.value, xs.value

    def export(self):
        m, n, nb, xs = self.dims()
        size = np.zeros(nb, np.int32); idx = np.zeros(nb, np.int32)
        x0 = np.zeros(xs); J0 = np.zeros(n * n); r0 = np.zeros(n)
        check(lib().tcv_prior_export(self.h, iptr(size), iptr(idx), dptr(x0), dptr(J0), dptr(r0)))
        xs_l, o = [], 0
        for s in size:
            xs_l.append(x0[o:o + s].copy()); o += int(s)
        return dict(m=m, n=n, sizes=[int(s) for s in size], idx=[int(i) - m for i in idx], x0=xs_l,
                    J0=J0.reshape(n, n).T.copy(), r0=r0)

    def schur(self):
        m, n, nb, xs = self.dims()
        As = np.zeros(n * n); bs = np.zeros(n)
        check(lib().tcv_prior_export_schur(self.h, dptr(As), dptr(bs)))
        return As.reshape(n, n), bs

    def __del__(self):
        if self.h is not None and _lib is not None:
            _lib.tcv_prior_destroy(self.h)
            self.h = None


_KIND = {"pose": 0, "sb": 1, "ex": 2, "td": 3}


class Window:
    """Caller-side state arrays of one sliding window (what Estimator owns, estimator.h:166-172)
    plus the problem handle built from them with tcv_problem_from_window."""

    def __init__(self, win: dict, estimate_extrinsic=True, prior: Prior | None = None, share: "Window | None" = None, imu_device=None):
        """imu_device: optional list of `Preint` handles (device-resident pre-integrations, `preintegrate_device`), one per IMU factor of the
        window (entries may be None): they replace the host constants of win["imu"]"""
        self.win = win
        self.imu_device = imu_device
        if share is not None:      # same caller-owned state arrays (parameter blocks are identified by address)
            self.pose, self.sb, self.ex, self.lam, self.td = share.pose, share.sb, share.ex, share.lam, share.td
        else:
            self.td = None if win.get("td") is None else f64([float(win["td"])])      # para_Td[0] (ESTIMATE_TD)
            self.pose = f64(win["pose"]).copy(); self.sb = f64(win["speedbias"]).copy()
            self.ex = f64(win["ex_pose"]).copy(); self.lam = f64(win["lam"]).copy()
        im, pr, ln = win["imu"], win["proj"], win["line"]
        self._imu = pack_imu(im)
        self._imu_i = i32(im["frame_i"]); self._imu_j = i32(im["frame_j"])
        self._pi = i32(pr["frame_i"]); self._pj = i32(pr["frame_j"]); self._pl = i32(pr["landmark"])
        self._pts = f64(np.concatenate([pr["pts_i"], pr["pts_j"]], -1)) if len(self._pi) else np.zeros((0, 6))
        self._lf = i32(ln["frame"])
        self._ld = f64(np.concatenate([ln["pts_start"], ln["pts_end"], ln["abc"]], -1)) if len(self._lf) else np.zeros((0, 9))
        d = WindowDesc()
        d.n_frames = self.pose.shape[0]; d.n_landmarks = self.lam.shape[0]
        d.n_imu = len(self._imu_i); d.n_proj = len(self._pi); d.n_line = len(self._lf)
        d.estimate_extrinsic = int(estimate_extrinsic)
        d.para_pose = dptr(self.pose); d.para_speedbias = dptr(self.sb); d.para_ex_pose = dptr(self.ex); d.para_feature = dptr(self.lam)
        d.imu_frame_i = iptr(self._imu_i); d.imu_frame_j = iptr(self._imu_j); d.imu = self._imu
        d.proj_frame_i = iptr(self._pi); d.proj_frame_j = iptr(self._pj); d.proj_feature = iptr(self._pl); d.proj_pts = dptr(self._pts)
        d.proj_sqrt_info = float(pr["sqrt_info"]); d.proj_loss_a = float(pr["loss_a"] or 0.0)
        d.line_frame = iptr(self._lf); d.line_data = dptr(self._ld)
        d.line_K[:] = list(np.asarray(ln["K"]).reshape(9)); d.line_Ric[:] = list(np.asarray(ln["Ric"]).reshape(9))
        d.line_Tic[:] = list(np.asarray(ln["Tic"]).reshape(3)); d.line_loss_a = float(ln["loss_a"] or 0.0)
        d.gravity[:] = list(np.asarray(win["G"]).reshape(3))
        d.line_exact_jacobian = int(bool(ln.get("exact_jacobian", False)))
        if self.td is not None:
            n = len(self._pi)
            self._aux = f64(np.concatenate([np.asarray(pr["vel_i"], dtype=float).reshape(n, 2), np.asarray(pr["vel_j"], dtype=float).reshape(n, 2),
                                            np.stack([np.asarray(pr[k], dtype=float).reshape(n) for k in ("td_i", "td_j", "row_i", "row_j")], -1)], -1)) if n else np.zeros((0, 8))
            d.para_td = dptr(self.td); d.proj_td_aux = dptr(self._aux); d.td_TR = float(pr["TR"]); d.td_ROW = float(pr["ROW"])
        if imu_device is not None:
            self._imu_dev = (C.c_void_p * len(self._imu_i))(*[(h.h if h is not None else None) for h in imu_device])
            d.imu_device = C.cast(self._imu_dev, C.POINTER(C.c_void_p))
        self.prior = prior
        if prior is None and win.get("prior") is not None:
            self.prior = Prior.from_dict(win["prior"])
        if self.prior is not None:
            blocks = win["prior"]["blocks"] if win.get("prior") is not None else getattr(self, "prior_blocks", [])      # (an empty prior, n = 0, has none)
            self._pk = i32([_KIND[b[0]] for b in blocks]); self._pidx = i32([b[1] for b in blocks])
            d.prior = self.prior.h; d.prior_block_kind = iptr(self._pk); d.prior_block_index = iptr(self._pidx)
        self.desc = d
        self.h = C.c_void_p()
        check(lib().tcv_problem_from_window(C.byref(d), C.byref(self.h)))

    def set_prior(self, prior: "Prior"):
        """tcv_problem_set_marginalization_prior: hand the problem a new prior with the layout of its current one (the same blocks)"""
        check(lib().tcv_problem_set_marginalization_prior(self.h, prior.h))
        self.prior = prior

    def states(self):
        out = dict(pose=self.pose.copy(), sb=self.sb.copy(), ex=self.ex.copy(), lam=self.lam.copy())
        if self.td is not None:
            out["td"] = self.td.copy()
        return out

    def plan_stats(self):
        out = np.zeros(16, np.int32)
        check(lib().tcv_problem_plan_stats(self.h, iptr(out)))
        keys = ["nc", "nx", "npp", "nland", "nt", "n_vis_chunk", "n_imu_chunk", "n_vunit", "n_vitem", "n_sunit", "n_sitem",
                "n_iunit", "n_iitem", "plan_ints", "window_doubles", "lds_bytes"]
        return dict(zip(keys, [int(v) for v in out]))

    def block_ptr(self, name, i):
        if name == "pose":
            return C.cast(C.addressof(self.pose.ctypes.data_as(_dp).contents) + 56 * i, _dp)
        if name == "sb":
            return C.cast(C.addressof(self.sb.ctypes.data_as(_dp).contents) + 72 * i, _dp)
        if name == "ex":
            return dptr(self.ex)
        if name == "td":
            return dptr(self.td)
        return C.cast(C.addressof(self.lam.ctypes.data_as(_dp).contents) + 8 * i, _dp)

    def __del__(self):
        if getattr(self, "h", None) is not None and _lib is not None:
            _lib.tcv_problem_destroy(self.h)
            self.h = None


def margin_old_window(win: dict) -> dict:
    """Factor set Estimator::OptimizationWithLine hands to MarginalizationInfo when the oldest frame is
    marginalised (estimator.cpp:1911-1986): the prior, the IMU factor (0,1) and every projection factor
    anchored in frame 0.  Line factors are not added (`if (0)`, :1992)."""
    im, pr, ln = win["imu"], win["proj"], win["line"]
    ki = [k for k in range(len(im["frame_i"])) if int(im["frame_i"][k]) == 0 and float(im["sum_dt"][k]) < 10.0]
    kp = [k for k in range(len(pr["frame_i"])) if int(pr["frame_i"][k]) == 0]
    out = dict(win)
    out["imu"] = {k: (np.asarray(v)[ki] if isinstance(v, np.ndarray) and v.shape[:1] == (len(im["frame_i"]),) else v) for k, v in im.items()}
    out["proj"] = {k: (np.asarray(v)[kp] if isinstance(v, np.ndarray) and v.shape[:1] == (len(pr["frame_i"]),) else v) for k, v in pr.items()}
    out["line"] = dict(ln, frame=np.zeros(0, int), pts_start=np.zeros((0, 3)), pts_end=np.zeros((0, 3)), abc=np.zeros((0, 3)))
    return out


def margin_second_new_window(win: dict) -> dict:
    """MARGIN_SECOND_NEW (estimator.cpp:2047-2113): MarginalizationInfo receives only the old prior."""
    im, pr, ln = win["imu"], win["proj"], win["line"]
    none = np.zeros(0, int)
    out = dict(win)
    out["imu"] = {k: (np.asarray(v)[none] if isinstance(v, np.ndarray) and v.shape[:1] == (len(im["frame_i"]),) else v) for k, v in im.items()}
    out["proj"] = {k: (np.asarray(v)[none] if isinstance(v, np.ndarray) and v.shape[:1] == (len(pr["frame_i"]),) else v) for k, v in pr.items()}
    out["line"] = dict(ln, frame=np.zeros(0, int), pts_start=np.zeros((0, 3)), pts_end=np.zeros((0, 3)), abc=np.zeros((0, 3)))
    return out


def margin_second_new_drops(w: "Window"):
    """drop set of estimator.cpp:2057-2063: para_Pose[WINDOW_SIZE - 1]."""
    return [w.block_ptr("pose", w.pose.shape[0] - 2)]


def margin_old_drops(w: "Window", mwin: dict):
    """drop sets of estimator.cpp:1918-1924 (prior), :1939-1941 (IMU) and :1983-1985 (points): pose 0, speed-bias 0
    and the inverse depths of the landmarks anchored in frame 0."""
    drops = [w.block_ptr("pose", 0), w.block_ptr("sb", 0)]
    for l in sorted(set(int(v) for v in mwin["proj"]["landmark"])):
        drops.append(w.block_ptr("lam", l))
    return drops


def prior_blocks(prior: "Prior", w: "Window", shift):
    """getParameterBlocks(addr_shift) (marginalization_factor.cpp:301-321): the kept blocks as (name, index) after applying
    `shift(name, index)` -- the addr_shift map of estimator.cpp:2027-2039 (MARGIN_OLD) or :2084-2104 (MARGIN_SECOND_NEW)."""
    m, n, nb, xs = prior.dims()
    addrs = (_dp * nb)()
    check(lib().tcv_prior_keep_block_addresses(prior.h, addrs))
    base = {"pose": (w.pose.ctypes.data, 56, w.pose.shape[0]), "sb": (w.sb.ctypes.data, 72, w.sb.shape[0]), "ex": (w.ex.ctypes.data, 56, 1)}
    if getattr(w, "td", None) is not None:
        base["td"] = (w.td.ctypes.data, 8, 1)
    out = []
    for k in range(nb):
        a = C.cast(addrs[k], C.c_void_p).value
        for name, (b0, stride, cnt) in base.items():
            if b0 <= a < b0 + stride * cnt and (a - b0) % stride == 0:
                out.append(tuple(shift(name, (a - b0) // stride)))
                break
        else:
            raise ValueError("kept block is not a pose / speed-bias / extrinsic / td block")
    return out


def shifted_prior_blocks(prior: "Prior", w: "Window"):
    """getParameterBlocks(addr_shift) for MARGIN_OLD (estimator.cpp:2027-2039): pose i -> i-1, speed-bias i -> i-1."""
    return prior_blocks(prior, w, lambda name, i: (name, i - 1) if name in ("pose", "sb") else (name, 0))


class BatchSpec:
    """the argument arrays of tcv_batch_create (problem handles, marginalisation problem handles, drop lists) as C arrays"""

    def __init__(self, windows, marg_windows=None, marg_drops=None):
        self.windows = list(windows)
        n = self.n = len(self.windows)
        self.arr = (C.c_void_p * n)(*[w.h for w in self.windows])
        self.marg_windows, self.marr, self.dd, self.nd = None, None, None, None
        if marg_windows is not None:
            self.marg_windows = list(marg_windows)
            self.marr = (C.c_void_p * n)(*[(w.h if w is not None else None) for w in self.marg_windows])      # None: that window is only solved
            self._drop_arrays = []
            self.dd = (C.POINTER(_dp) * n)()
            self.nd = (C.c_int * n)()
            for k, drops in enumerate(marg_drops):
                drops = drops or []
                a = (_dp * max(1, len(drops)))(*drops)
                self._drop_arrays.append(a)
                self.dd[k] = C.cast(a, C.POINTER(_dp))
                self.nd[k] = len(drops)


class Batch:
    """Device-resident batch of independent windows (throughput mode)."""

    def __init__(self, windows, marg_windows=None, marg_drops=None, spec=None):
        """spec: a BatchSpec (the C arrays of tcv_batch_create, built once) instead of the three lists -- what a C / C++ caller passes anyway"""
        if spec is None:
            spec = BatchSpec(windows, marg_windows, marg_drops)
        self.spec = spec
        self.windows = spec.windows
        if spec.marg_windows is not None:
            self.marg_windows = spec.marg_windows
        self.h = C.c_void_p()
        check(lib().tcv_batch_create(C.byref(self.h), spec.arr, spec.marr, spec.dd, spec.nd, spec.n))

    def attach_marginalization(self, marg_windows, marg_drops):
        """tcv_batch_attach_marginalization: the marginalisation problems of a batch created without them (may overlap the batch's solve)"""
        sp = BatchSpec(self.windows, marg_windows, marg_drops)
        self._attached = sp
        self.marg_windows = sp.marg_windows
        check(lib().tcv_batch_attach_marginalization(self.h, sp.marr, sp.dd, sp.nd))

    def priors(self):
        """every window's prior in one call (tcv_batch_get_priors)"""
        n = len(self.windows)
        out = (C.c_void_p * n)()
        check(lib().tcv_batch_get_priors(self.h, out, n))
        return [Prior(C.c_void_p(h)) if h else None for h in out]

    def priors_device_raw(self):
        """tcv_batch_get_priors_device as a C array of handles (owned by the caller: tcv_priors_destroy) -- for loops that hand the priors
        of a whole batch on with `tcv_problems_set_marginalization_prior` instead of wrapping every handle in a Python object"""
        n = len(self.windows)
        out = (C.c_void_p * n)()
        check(lib().tcv_batch_get_priors_device(self.h, out, n))
        return out

    def priors_device(self, nowait=False):
        """every window's prior as a DEVICE-RESIDENT handle (tcv_batch_get_priors_device): layout on the host, J0 | r0 | x0 left in HBM; a
        problem that holds one uploads nothing of it.  nowait: tcv_batch_get_priors_device_async -- the marginalisation may still be running,
        consumers are ordered behind it on the device, its status is asked for later (marg_status)"""
        n = len(self.windows)
        out = (C.c_void_p * n)()
        check((lib().tcv_batch_get_priors_device_async if nowait else lib().tcv_batch_get_priors_device)(self.h, out, n))
        return [Prior(C.c_void_p(h)) if h else None for h in out]

    def solve(self, opts, stream=None):
        check(lib().tcv_batch_solve(self.h, C.byref(opts), stream))

    def marginalize(self, stream=None):
        check(lib().tcv_batch_marginalize(self.h, stream))

    def marg_status(self):
        out = np.zeros(len(self.windows), np.int32)
        check(lib().tcv_batch_marg_status(self.h, iptr(out), len(out)))
        return out

    def gauge_fix(self, stream=None):
        """Estimator::double2vector() + vector2double() in place on the solved states in HBM (estimator.cpp:1537-1581)."""
        check(lib().tcv_batch_gauge_fix(self.h, stream))

    def fuse_gauge_fix(self, on=True):
        """tcv_batch_set_fused_gauge_fix: the same fix in the solve kernel's epilogue; `gauge_fix()` behind such a solve is a no-op."""
        check(lib().tcv_batch_set_fused_gauge_fix(self.h, int(on)))

    def synchronize(self):
        check(lib().tcv_batch_synchronize(self.h))

    def download_states(self):
        check(lib().tcv_batch_download_states(self.h))

    def summaries(self, n=None):
        n = len(self.windows) if n is None else n
        arr = (SolverSummary * n)()
        check(lib().tcv_batch_get_summaries(self.h, arr, n))
        return arr

    def first_step(self, window):
        out = np.zeros(2048)
        ln = C.c_int()
        check(lib().tcv_batch_get_first_step(self.h, window, dptr(out), 2048, C.byref(ln)))
        return out[:ln.value].copy()

    def prior(self, window):
        h = C.c_void_p()
        check(lib().tcv_batch_get_prior(self.h, window, C.byref(h)))
        return Prior(h)

    def download_priors(self, compact=False):
        """one D2H copy of every window's marginalisation result; `prior(k)` is then served from the host copy.  compact: without A', b'
        (`Prior.schur()` is then unavailable)."""
        check((lib().tcv_batch_download_priors_compact if compact else lib().tcv_batch_download_priors)(self.h))

    def stats(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        check(lib().tcv_batch_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(input_bytes=a.value, solve_ms=b.value, marg_ms=c.value)

    def plan_stats(self):
        a, b, c, d = C.c_int(), C.c_double(), C.c_int(), C.c_int()
        check(lib().tcv_batch_plan_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(num_plans=a.value, plan_bytes=b.value, grid=c.value, lds_bytes=d.value, layout=("chain", "dense")[lib().tcv_batch_layout(self.h)])

    def cooperative(self):
        a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(lib().tcv_batch_cooperative(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(helpers=a.value, groups=b.value, chunks=c.value, last_solve_workgroups=d.value)

    def __del__(self):
        if getattr(self, "h", None) is not None and _lib is not None:
            _lib.tcv_batch_destroy(self.h)
            self.h = None


# ---- batched factor evaluation (CostFunction::Evaluate layout) ---------------------------------------
def eval_imu(imu, params, G, sqrt_info=None, want_jac=True):
    n = len(imu["frame_i"])
    pre = pack_imu(imu)
    params = f64(params).reshape(n, 32)
    S = np.zeros((n, 225)) if sqrt_info is None else f64(sqrt_info).reshape(n, 225).copy()
    res = np.zeros((n, 15)); jac = np.zeros((n, 480))
    Gv = f64(G)
    check(lib().tcv_eval_imu_factors(n, pre, dptr(params), dptr(Gv), int(sqrt_info is not None), dptr(S), dptr(res),
                                     dptr(jac) if want_jac else None))
    Js = [jac[:, 0:105].reshape(n, 15, 7), jac[:, 105:240].reshape(n, 15, 9), jac[:, 240:345].reshape(n, 15, 7),
          jac[:, 345:480].reshape(n, 15, 9)]
    return res, Js, S.reshape(n, 15, 15)


def eval_proj(pts, params, sqrt_info, want_jac=True):
    pts = f64(pts); params = f64(params)
    n = pts.shape[0]
    res = np.zeros((n, 2)); jac = np.zeros((n, 44))
    check(lib().tcv_eval_projection_factors(n, dptr(pts), dptr(params), float(sqrt_info), dptr(res), dptr(jac) if want_jac else None))
    return res, [jac[:, 0:14].reshape(n, 2, 7), jac[:, 14:28].reshape(n, 2, 7), jac[:, 28:42].reshape(n, 2, 7),
                 jac[:, 42:44].reshape(n, 2, 1)]


def eval_proj_td(pts, aux, params, sqrt_info, TR, ROW, want_jac=True):
    pts = f64(pts); aux = f64(aux); params = f64(params)
    n = pts.shape[0]
    res = np.zeros((n, 2)); jac = np.zeros((n, 46))
    check(lib().tcv_eval_projection_td_factors(n, dptr(pts), dptr(aux), dptr(params), float(sqrt_info), float(TR), float(ROW), dptr(res),
                                               dptr(jac) if want_jac else None))
    return res, [jac[:, 0:14].reshape(n, 2, 7), jac[:, 14:28].reshape(n, 2, 7), jac[:, 28:42].reshape(n, 2, 7),
                 jac[:, 42:44].reshape(n, 2, 1), jac[:, 44:46].reshape(n, 2, 1)]


def eval_line(line, K, R, T, params, want_jac=True):
    line = f64(line); params = f64(params)
    n = line.shape[0]
    res = np.zeros((n, 2)); jac = np.zeros((n, 14))
    K = f64(K).reshape(9); R = f64(R).reshape(9); T = f64(T).reshape(3)
    check(lib().tcv_eval_line_factors(n, dptr(line), dptr(K), dptr(R), dptr(T), dptr(params), dptr(res), dptr(jac) if want_jac else None))
    return res, jac.reshape(n, 2, 7)


def pose_plus(x, delta):
    x = f64(x); delta = f64(delta)
    n = x.shape[0]
    out = np.zeros((n, 7))
    check(lib().tcv_pose_plus(n, dptr(x), dptr(delta), dptr(out)))
    return out


def preintegrate(acc, gyr, dt, lin_ba, lin_bg, noise):
    """Batched IntegrationBase: acc/gyr (n, S+1, 3) with sample 0 = constructor's (acc_0, gyr_0), samples 1..S pushed with
    step dt; lin_ba/lin_bg (n, 3); noise = (ACC_N, GYR_N, ACC_W, GYR_W).  Returns the dict layout of synth's 'imu'."""
    acc = f64(acc); gyr = f64(gyr)
    n, S = acc.shape[0], acc.shape[1] - 1
    samples = np.concatenate([np.full((n, S, 1), float(dt)), acc[:, 1:], gyr[:, 1:]], -1).reshape(n * S, 7).copy()
    first = i32(np.arange(n) * S); count = i32(np.full(n, S))
    init = f64(np.concatenate([acc[:, 0], gyr[:, 0], f64(lin_ba).reshape(n, 3), f64(lin_bg).reshape(n, 3)], -1))
    nz = f64(noise)
    out = (ImuPreintegration * n)()
    check(lib().tcv_preintegrate(n, iptr(first), iptr(count), dptr(samples), n * S, dptr(init), dptr(nz), out))
    return dict(delta_p=np.array([list(o.delta_p) for o in out]), delta_q=np.array([list(o.delta_q) for o in out]),
                delta_v=np.array([list(o.delta_v) for o in out]), lin_ba=np.array([list(o.linearized_ba) for o in out]),
                lin_bg=np.array([list(o.linearized_bg) for o in out]), sum_dt=np.array([o.sum_dt for o in out]),
                jacobian=np.array([list(o.jacobian) for o in out]).reshape(n, 15, 15),
                covariance=np.array([list(o.covariance) for o in out]).reshape(n, 15, 15))


class MatchLinesArgs(C.Structure):
    _fields_ = [("n_frames", C.c_int), ("poses", _dp), ("ex_pose", _dp), ("Rbw", _dp), ("Tbw", _dp), ("K", _dp), ("width", C.c_int), ("height", C.c_int),
                ("window_size", C.c_int), ("n_map", C.c_int), ("lines3d", _dp), ("n_det", C.c_int), ("det_frame", _ip), ("det_lines", _dp),
                ("angle_th", C.c_double), ("overlap_th", C.c_double), ("fov_given", C.c_int), ("in_fov", C.POINTER(C.c_ubyte)),
                ("match_index", _ip), ("err", C.POINTER(C.c_float)), ("projected", _dp), ("map_device", C.c_void_p)]


def match_lines_batch(calls):
    """tcv_match_lines_batch: `calls` = list of dicts with the keyword arguments of `match_lines`; ONE device round trip for all of them.
    Returns a list of (in_fov, match_index, err, projected) tuples."""
    n = len(calls)
    arr = (MatchLinesArgs * n)()
    keep, outs = [], []
    for k, c in enumerate(calls):
        poses = f64(c["poses"]).reshape(-1, 7); lines3d = f64(c["lines3d"]).reshape(-1, 6)
        det_frame = i32(c["det_frame"]); det_lines = f64(c["det_lines"]).reshape(-1, 4)
        nf, nm, nd = poses.shape[0], lines3d.shape[0], det_lines.shape[0]
        in_fov, fov_frame = c.get("in_fov"), c.get("fov_frame")
        fov = np.zeros((nf, nm), np.uint8) if in_fov is None else np.ascontiguousarray(np.asarray(in_fov).reshape(nf, nm), dtype=np.uint8)
        match = np.zeros(max(nd, 1), np.int32); err = np.zeros((max(nd, 1), 3), np.float32); proj = np.zeros((max(nd, 1), 4))
        ex = f64(c["ex_pose"]); R = f64(c["Rbw"]).reshape(9); T = f64(c["Tbw"]); Kf = f64(c["K"]).reshape(9)
        a = arr[k]
        a.n_frames = nf; a.poses = dptr(poses); a.ex_pose = dptr(ex); a.Rbw = dptr(R); a.Tbw = dptr(T); a.K = dptr(Kf)
        a.width = int(c["width"]); a.height = int(c["height"]); a.window_size = int(c["window_size"]); a.n_map = nm; a.lines3d = dptr(lines3d)
        a.n_det = nd; a.det_frame = iptr(det_frame) if nd else None; a.det_lines = dptr(det_lines) if nd else None
        a.angle_th = float(c["angle_th"]); a.overlap_th = float(c["overlap_th"])
        a.fov_given = (2 + int(fov_frame)) if (in_fov is not None and fov_frame is not None) else int(in_fov is not None)
        a.in_fov = fov.ctypes.data_as(C.POINTER(C.c_ubyte)); a.match_index = iptr(match); a.err = err.ctypes.data_as(C.POINTER(C.c_float)); a.projected = dptr(proj)
        keep.append((poses, lines3d, det_frame, det_lines, ex, R, T, Kf))
        outs.append((fov, match, err, proj, nd))
    lib().tcv_match_lines_batch.argtypes = [C.c_int, C.POINTER(MatchLinesArgs)]
    check(lib().tcv_match_lines_batch(n, arr))
    return [(fov.astype(bool), match[:nd].copy(), err[:nd].copy(), proj[:nd].copy()) for fov, match, err, proj, nd in outs]


class Preint:
    """Owns a tcv_preint handle: an IntegrationBase whose numbers stay on the device."""

    def __init__(self, h):
        self.h = h

    def sum_dt(self):
        return float(lib().tcv_preint_sum_dt(self.h))

    def export(self):
        o = ImuPreintegration()
        check(lib().tcv_preint_export(self.h, C.byref(o)))
        return dict(delta_p=np.array(o.delta_p), delta_q=np.array(o.delta_q), delta_v=np.array(o.delta_v), lin_ba=np.array(o.linearized_ba),
                    lin_bg=np.array(o.linearized_bg), sum_dt=float(o.sum_dt), jacobian=np.array(o.jacobian).reshape(15, 15), covariance=np.array(o.covariance).reshape(15, 15))

    def __del__(self):
        if self.h is not None and _lib is not None:
            _lib.tcv_preint_destroy(self.h)
            self.h = None


def preintegrate_device(acc, gyr, dt, lin_ba, lin_bg, noise):
    """`preintegrate` with the results left on the device: a list of `Preint` handles"""
    acc = f64(acc); gyr = f64(gyr)
    n, S = acc.shape[0], acc.shape[1] - 1
    samples = np.concatenate([np.full((n, S, 1), float(dt)), acc[:, 1:], gyr[:, 1:]], -1).reshape(n * S, 7).copy()
    first = i32(np.arange(n) * S); count = i32(np.full(n, S))
    init = f64(np.concatenate([acc[:, 0], gyr[:, 0], f64(lin_ba).reshape(n, 3), f64(lin_bg).reshape(n, 3)], -1))
    nz = f64(noise)
    out = (C.c_void_p * n)()
    check(lib().tcv_preintegrate_device(n, iptr(first), iptr(count), dptr(samples), n * S, dptr(init), dptr(nz), out))
    return [Preint(C.c_void_p(h)) for h in out]


def gauge_fix(R0, P0, pose, sb):
    """Estimator::double2vector() (estimator.cpp:1537-1581) on the GPU: returns Rs (n,3,3), Ps, Vs and the para_Pose
    the next vector2double() writes."""
    pose = f64(pose); sb = f64(sb); R0 = f64(R0).reshape(9); P0 = f64(P0)
    n = pose.shape[0]
    Rs = np.zeros((n, 3, 3)); Ps = np.zeros((n, 3)); Vs = np.zeros((n, 3)); po = np.zeros((n, 7))
    check(lib().tcv_gauge_fix(n, dptr(R0), dptr(P0), dptr(pose), dptr(sb), dptr(Rs), dptr(Ps), dptr(Vs), dptr(po)))
    return Rs, Ps, Vs, po


def match_lines(poses, ex_pose, Rbw, Tbw, K, width, height, window_size, lines3d, det_frame, det_lines, angle_th, overlap_th, in_fov=None, fov_frame=None):
    """UpdateLinesInFoV + LineCorrespondenceInFrame (estimator.cpp:385-447, :671-885) on the GPU.
    in_fov given: the matching runs against these frozen FoV sets; with fov_frame = f the row of frame f is computed first and returned.
    Returns (in_fov bool (n_frames, n_map), match_index (n_det,), err float32 (n_det, 3), projected (n_det, 4))."""
    poses = f64(poses).reshape(-1, 7); lines3d = f64(lines3d).reshape(-1, 6)
    det_frame = i32(det_frame); det_lines = f64(det_lines).reshape(-1, 4)
    nf, nm, nd = poses.shape[0], lines3d.shape[0], det_lines.shape[0]
    fov = np.zeros((nf, nm), np.uint8) if in_fov is None else np.ascontiguousarray(np.asarray(in_fov).reshape(nf, nm), dtype=np.uint8)
    match = np.zeros(max(nd, 1), np.int32); err = np.zeros((max(nd, 1), 3), np.float32); proj = np.zeros((max(nd, 1), 4))
    ex = f64(ex_pose); R = f64(Rbw).reshape(9); T = f64(Tbw); Kf = f64(K).reshape(9)
    check(lib().tcv_match_lines(nf, dptr(poses), dptr(ex), dptr(R), dptr(T), dptr(Kf), int(width), int(height), int(window_size), nm, dptr(lines3d),
                                nd, iptr(det_frame) if nd else None, dptr(det_lines) if nd else None, float(angle_th), float(overlap_th),
                                (2 + int(fov_frame)) if (in_fov is not None and fov_frame is not None) else int(in_fov is not None), fov.ctypes.data_as(C.POINTER(C.c_ubyte)), iptr(match), err.ctypes.data_as(C.POINTER(C.c_float)), dptr(proj)))
    return fov.astype(bool), match[:nd].copy(), err[:nd].copy(), proj[:nd].copy()
