"""GPU parity of the device IMU pre-integration (IntegrationBase, integration_base.h:13-158)
against the NumPy and C oracles on the synthetic 200 Hz IMU streams."""
import numpy as np
import pytest

import np_oracle as npo
import orc
import synth
from util import rel

pytestmark = pytest.mark.gpu
NOISE = (synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W)


def test_preintegration_matches_oracles(gpu):
    batch = synth.make_windows(1234, 3)
    acc = batch["imu"]["acc"].reshape(-1, synth.IMU_RATE_SUB + 1, 3); gyr = batch["imu"]["gyr"].reshape(-1, synth.IMU_RATE_SUB + 1, 3)
    n = acc.shape[0]
    rng = np.random.default_rng(5)
    ba = 0.02 * rng.normal(size=(n, 3)); bg = 0.002 * rng.normal(size=(n, 3))      # non-zero linearisation biases
    out = gpu.preintegrate(acc, gyr, synth.DT_IMU, ba, bg, NOISE)
    for k in range(n):
        ref = npo.preintegrate(acc[k], gyr[k], synth.DT_IMU, ba[k], bg[k], *NOISE)
        assert rel(out["delta_p"][k], ref["delta_p"]) < 1e-12 and rel(out["delta_v"][k], ref["delta_v"]) < 1e-12
        assert rel(out["delta_q"][k], ref["delta_q"]) < 1e-13
        assert abs(out["sum_dt"][k] - ref["sum_dt"]) < 1e-15
        assert rel(out["jacobian"][k], ref["jacobian"]) < 1e-11
        assert rel(out["covariance"][k], ref["covariance"]) < 1e-11
        assert np.array_equal(out["lin_ba"][k], ba[k]) and np.array_equal(out["lin_bg"][k], bg[k])
        assert abs(np.linalg.norm(out["delta_q"][k]) - 1) < 1e-14        # delta_q.normalize() every step (:153)
    # C oracle, packed 287-constant layout
    c = np.zeros(orc.IMU_STRIDE); jac = np.zeros(225)
    a0 = np.ascontiguousarray(acc[0]); g0 = np.ascontiguousarray(gyr[0])
    orc.lib().orc_preintegrate(orc.dptr(a0), orc.dptr(g0), synth.IMU_RATE_SUB + 1, synth.DT_IMU, orc.dptr(ba[0].copy()), orc.dptr(bg[0].copy()),
                               *NOISE, orc.dptr(c), orc.dptr(jac))
    assert rel(out["delta_p"][0], c[0:3]) < 1e-12 and rel(out["covariance"][0].reshape(225), c[62:287]) < 1e-11
    assert rel(out["jacobian"][0].reshape(225), jac) < 1e-11


def test_repropagate_and_generated_windows_agree(gpu):
    """synth's own host-side pre-integration (zero biases) is reproduced, and re-propagating with new biases
    (IntegrationBase::repropagate, :38-52) equals a fresh integration with those biases."""
    batch = synth.make_windows(77, 2)
    im = batch["imu"]
    acc = im["acc"].reshape(-1, synth.IMU_RATE_SUB + 1, 3); gyr = im["gyr"].reshape(-1, synth.IMU_RATE_SUB + 1, 3)
    n = acc.shape[0]
    z = np.zeros((n, 3))
    out = gpu.preintegrate(acc, gyr, synth.DT_IMU, z, z, NOISE)
    assert rel(out["covariance"], im["covariance"].reshape(n, 15, 15)) < 1e-11
    assert rel(out["jacobian"], im["jacobian"].reshape(n, 15, 15)) < 1e-11
    assert rel(out["delta_p"], im["delta_p"].reshape(n, 3)) < 1e-12
    ba = np.full((n, 3), 0.01); bg = np.full((n, 3), -0.001)
    rep = gpu.preintegrate(acc, gyr, synth.DT_IMU, ba, bg, NOISE)
    assert rel(rep["delta_v"], out["delta_v"]) > 1e-6                 # the biases matter
    # first-order bias correction of the factor (integration_base.h:173-178) predicts the re-propagated deltas
    J = out["jacobian"]
    pred_v = out["delta_v"] + np.einsum("nij,nj->ni", J[:, 6:9, 9:12], ba) + np.einsum("nij,nj->ni", J[:, 6:9, 12:15], bg)
    assert rel(pred_v, rep["delta_v"]) < 1e-4


def test_preintegrate_edge_cases(gpu):
    acc = np.tile(np.array([0.0, 0.0, 9.81]), (1, 2, 1)); gyr = np.zeros((1, 2, 3))
    out = gpu.preintegrate(acc, gyr, 0.005, np.zeros((1, 3)), np.zeros((1, 3)), NOISE)      # a single sample
    assert abs(out["sum_dt"][0] - 0.005) < 1e-18 and rel(out["delta_v"][0], [0, 0, 9.81 * 0.005]) < 1e-14
    assert np.allclose(out["delta_q"][0], [0, 0, 0, 1])
    out0 = gpu.preintegrate(acc[:, :1], gyr[:, :1], 0.005, np.zeros((1, 3)), np.zeros((1, 3)), NOISE)  # zero samples: constructor state
    assert out0["sum_dt"][0] == 0.0 and np.array_equal(out0["jacobian"][0], np.eye(15)) and not out0["covariance"][0].any()
