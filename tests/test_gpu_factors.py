"""GPU parity of the residual/Jacobian evaluators (called through the C-ABI, CostFunction::Evaluate layout)
against the committed golden vectors and the C oracle."""
import numpy as np
import pytest

import orc
from util import imu_pre, load, rel

pytestmark = pytest.mark.gpu
TOL = 1e-10          # FP64 re-association only (FMA contraction on the GPU)


def test_projection_factor_golden(gpu):
    z = load("factors.npz")
    n = len(z["p1_lam"])
    params = np.concatenate([z["p1_pose_i"], z["p1_pose_j"], z["p1_ex"], z["p1_lam"][:, None]], 1)
    pts = np.concatenate([z["p1_pts_i"], z["p1_pts_j"]], 1)
    r, Js = gpu.eval_proj(pts, params, float(z["p1_sqrt_info"]))
    assert rel(r, z["p1_r"]) < TOL
    for b in range(4):
        assert rel(Js[b], z[f"p1_J{b}"]) < TOL
    for b in range(3):
        assert np.all(Js[b][:, :, 6] == 0.0)
    r2, _ = gpu.eval_proj(pts, params, float(z["p1_sqrt_info"]), want_jac=False)     # jacobians == NULL path
    assert np.array_equal(r, r2)


def test_line_factor_golden_including_as_written_jacobian(gpu):
    z = load("factors.npz")
    line = np.concatenate([z["l1_start"], z["l1_end"], z["l1_abc"]], 1)
    r, J = gpu.eval_line(line, z["l1_K"], z["l1_Ric"], z["l1_Tic"], z["l1_pose"])
    assert rel(r, z["l1_r"]) < TOL and rel(J, z["l1_J"]) < TOL
    assert np.all(J[:, :, 6] == 0.0) and np.all(r >= 0.0)      # residual = point-to-line distances (line_projection_factor.cpp:68-69)


def test_imu_factor_golden_given_sqrt_info(gpu):
    z = load("factors.npz")
    n = len(z["i1_sum_dt"])
    imu = {k: z["i1_" + k] for k in ("delta_p", "delta_q", "delta_v", "lin_ba", "lin_bg", "sum_dt", "jacobian", "covariance")}
    imu["frame_i"] = np.zeros(n, int)
    params = np.concatenate([z["i1_pose_i"], z["i1_sb_i"], z["i1_pose_j"], z["i1_sb_j"]], 1)
    r, Js, _ = gpu.eval_imu(imu, params, z["i1_G"], sqrt_info=z["i1_sqrt_info"])
    assert rel(r, z["i1_r"]) < 1e-9
    for b in range(4):
        assert rel(Js[b], z[f"i1_J{b}"]) < 1e-9
    assert np.all(Js[0][:, :, 6] == 0.0) and np.all(Js[2][:, :, 6] == 0.0)


def test_imu_sqrt_info_on_device(gpu):
    """imu_factor.h:64 evaluated in the kernel: same operation order as the C oracle, FMA contraction off."""
    z = load("factors.npz")
    n = len(z["i1_sum_dt"])
    imu = {k: z["i1_" + k] for k in ("delta_p", "delta_q", "delta_v", "lin_ba", "lin_bg", "sum_dt", "jacobian", "covariance")}
    imu["frame_i"] = np.zeros(n, int)
    params = np.concatenate([z["i1_pose_i"], z["i1_sb_i"], z["i1_pose_j"], z["i1_sb_j"]], 1)
    r, Js, S = gpu.eval_imu(imu, params, z["i1_G"])
    for k in range(n):
        S_c = orc.imu_sqrt_info(z["i1_covariance"][k])
        assert rel(S[k], S_c) < 1e-9
        assert np.all(np.tril(S[k], -1) == 0.0)                 # matrixL().transpose() is upper triangular
        assert rel(S[k].T @ S[k], np.linalg.inv(z["i1_covariance"][k])) < 1e-6


def test_pose_plus_golden_and_edge_cases(gpu):
    z = load("factors.npz")
    out = gpu.pose_plus(z["s2_x"], z["s2_delta"])
    assert rel(out, z["s2_out"]) < 1e-14
    assert np.abs(np.linalg.norm(out[:, 3:], axis=1) - 1).max() < 1e-14
    big = gpu.pose_plus(z["s2_x"][:2], np.array([[0, 0, 0, 3.0, -2.0, 1.0], [1e3, -1e3, 0, 0, 0, 0]]))   # large first-order rotation
    assert np.all(np.isfinite(big))


def test_single_factor_and_large_batch(gpu):
    z = load("factors.npz")
    params = np.concatenate([z["p1_pose_i"], z["p1_pose_j"], z["p1_ex"], z["p1_lam"][:, None]], 1)
    pts = np.concatenate([z["p1_pts_i"], z["p1_pts_j"]], 1)
    r1, _ = gpu.eval_proj(pts[:1], params[:1], float(z["p1_sqrt_info"]))
    assert rel(r1[0], z["p1_r"][0]) < TOL
    reps = 4099                                                   # not a multiple of the block size
    idx = np.arange(reps) % len(pts)
    r, Js = gpu.eval_proj(pts[idx], params[idx], float(z["p1_sqrt_info"]))
    assert rel(r, z["p1_r"][idx]) < TOL and rel(Js[2], z["p1_J2"][idx]) < TOL


def test_projection_td_factor_golden(gpu):
    """T1 ProjectionTdFactor::Evaluate (projection_td_factor.cpp:34-140) on the device vs the golden vectors."""
    z = load("proj_td.npz")
    for tr in (0.03, 0.0):
        sel = np.nonzero(z["TR"] == tr)[0]
        res, Js = gpu.eval_proj_td(z["pts"][sel], z["aux"][sel], z["params"][sel], float(z["sqrt_info"]), tr, float(z["ROW"]))
        assert rel(res, z["res"][sel]) < 1e-10
        for J, nm in zip(Js, ["J_pose_i", "J_pose_j", "J_ex", "J_lam", "J_td"]):
            assert rel(J, z[nm][sel]) < 1e-10, nm
            if J.shape[-1] == 7:
                assert np.all(J[..., 6] == 0.0)
    res2, _ = gpu.eval_proj_td(z["pts"][:3], z["aux"][:3], z["params"][:3], float(z["sqrt_info"]), 0.03, float(z["ROW"]), want_jac=False)
    assert np.array_equal(res2, gpu.eval_proj_td(z["pts"][:3], z["aux"][:3], z["params"][:3], float(z["sqrt_info"]), 0.03, float(z["ROW"]))[0])
