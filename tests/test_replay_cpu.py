"""N1 / N2 on the CPU: the replay harness's window management driven by the oracle back end, the
trajectory CSV format and the ATE evaluator."""
import os

import numpy as np

import ate
import replay
from replay_oracle import OracleBackend


def test_umeyama_recovers_a_known_similarity_and_ate_is_zero():
    rng = np.random.default_rng(3)
    p = rng.normal(size=(50, 3))
    R = replay.q2R(np.array([0.1, -0.2, 0.3, 0.9]) / np.linalg.norm([0.1, -0.2, 0.3, 0.9])); t = np.array([1.0, -2.0, 0.5])
    s, R2, t2 = ate.umeyama(p, 1.7 * p @ R.T + t, with_scale=True)
    assert abs(s - 1.7) < 1e-12 and np.allclose(R2, R, atol=1e-12) and np.allclose(t2, t, atol=1e-12)
    assert ate.ate_rmse(p, p @ R.T + t) < 1e-12
    assert abs(ate.ate_rmse(p, p + np.array([0.0, 0.0, 0.003]), align=False) - 0.003) < 1e-15
    noisy = p @ R.T + t + rng.normal(size=p.shape) * 0.01
    assert 0.005 < ate.ate_rmse(p, noisy) < 0.03


def test_vins_result_csv_round_trip(tmp_path):
    f = os.path.join(tmp_path, "vins_result.csv")
    t = np.array([1403715273.262142976, 1403715273.362142976])
    P = np.array([[1.0, 2.0, 3.0], [1.1, 2.1, 3.1]]); q = np.array([[0.0, 0.0, 0.0, 1.0], [0.1, 0.2, 0.3, 0.9273618495495704]]); V = np.zeros((2, 3))
    ate.write_vins_result(f, t, P, q, V)
    first = open(f).readline()
    assert first == "1403715273262142976,1.00000,2.00000,3.00000,1.00000,0.00000,0.00000,0.00000,0.00000,0.00000,0.00000,\n"   # visualization.cpp:211-226
    d = ate.read_vins_result(f)
    assert np.allclose(d["p"], P) and np.allclose(d["q_xyzw"], q, atol=1e-5) and np.allclose(d["t"], t, atol=1e-6)
    i, j = ate.associate(d["t"], t + 0.001)
    assert list(i) == [0, 1] and list(j) == [0, 1]


def test_replay_with_the_oracle_back_end_tracks_the_ground_truth():
    """40 frames: both marginalisation modes occur, priors chain, features are triangulated / slid / dropped.  With exact
    data the replay reproduces the trajectory to 1e-5 m; with 1 px / IMU noise and the weak excitation of the synthetic
    platform (1 s windows, tilt <-> accelerometer-bias ambiguity) the SHAPE stays within a few cm (aligned ATE) while yaw and
    position of the window drift, as they are unobservable."""
    clean = replay.simulate_stream(1, 24, max_features=30, pace=0.0, pixel_sigma=0.0, imu_noise=False)
    for k in range(1, len(clean["imu"])):
        a, g = clean["imu"][k]; clean["imu"][k] = (a - clean["ba"], g - clean["bg"])
    clean["ba"] = np.zeros(3); clean["bg"] = np.zeros(3)
    out = replay.run(clean, OracleBackend(), num_iterations=8, init_sigma=(0, 0, 0), bias_sigma=(0, 0))
    i, j = ate.associate(out["t"], clean["t"])
    assert ate.ate_rmse(out["p"][i], clean["gt_p"][j], align=False) < 1e-4
    assert max(l["final_cost"] for l in out["log"]) < 1e-2

    stream = replay.simulate_stream(1, 40, max_features=30)
    out = replay.run(stream, OracleBackend(), num_iterations=8)
    assert len(out["t"]) == 40 - replay.WINDOW_SIZE
    flags = [l["flag"] for l in out["log"]]
    assert replay.MARGIN_OLD in flags and replay.MARGIN_SECOND_NEW in flags
    assert all(l["prior_n"] is not None and l["prior_n"] <= 75 for l in out["log"][1:])
    i, j = ate.associate(out["t"], stream["t"])
    assert len(i) == len(out["t"])
    assert ate.ate_rmse(out["p"][i], stream["gt_p"][j]) < 0.10                    # SE(3)-aligned ATE
    assert ate.ate_rmse(out["p"][i], stream["gt_p"][j], align=False) < 1.0       # no divergence


def test_replay_with_line_association_in_the_loop():
    """N4 in the loop: the stream carries un-associated line tracks + the prior map; every frame runs UpdateLinesInFoV,
    updateLinePairInWindow / LineCorrespondenceInFrame and removeLineOutlier before the solve (processImagewithLine :328-336)."""
    stream = replay.simulate_stream(1, 30, max_features=30, associate=True)
    out = replay.run(stream, OracleBackend(), num_iterations=8)
    assert all(0 < l["n_line"] <= l["n_line_obs"] for l in out["log"])
    assert any(l["n_line"] < l["n_line_obs"] for l in out["log"])          # some observations fail the angle / overlap / distance gates
    i, j = ate.associate(out["t"], stream["t"])
    assert ate.ate_rmse(out["p"][i], stream["gt_p"][j]) < 0.10


def test_lock_step_multi_sequence_replay_equals_individual_replays():
    """run_many (all sequences' windows as one batch per frame, one pre-integration call) vs run per sequence: same results."""
    streams = [replay.simulate_stream(10 + k, 22, max_features=24) for k in range(3)]
    many = replay.run_many(streams, OracleBackend(), num_iterations=4)
    for st, m in zip(streams, many):
        one = replay.run(st, OracleBackend(), num_iterations=4)
        assert np.array_equal(one["p"], m["p"]) and np.array_equal(one["q"], m["q"]) and [l["flag"] for l in one["log"]] == [l["flag"] for l in m["log"]]


def test_euroc_excerpts_and_the_synthesised_imu_are_consistent_with_the_ground_truth():
    """N1's stream simulator on the EuRoC ground truth the reference ships (benchmark_publisher/config/<seq>/data.csv excerpts
    in tc-viml_amd/data/): every sequence loads, carries its own prior line map and map -> world transform, and the IMU samples
    synthesised from the ground truth integrate back onto it (mid-point integration as in Estimator::processIMU, open loop, 5 s:
    a few millimetres)."""
    assert len(replay.EUROC_SEQUENCES) == 5
    for seq in replay.EUROC_SEQUENCES:
        E = replay.load_euroc(seq)
        assert E["p"].shape == (7200, 3) and E["lines3d"].shape[1] == 6 and E["lines3d"].shape[0] in (891, 908)
        assert np.abs(np.diff(E["t"]) - 0.005).max() < 1e-6
        assert np.abs(E["Rbw"] @ E["Rbw"].T - np.eye(3)).max() < 1e-5
    a = replay.simulate_stream_euroc("V2_02_medium", 51, start_s=3.0, imu_noise=False)
    assert a["stamp_ns"][0] > 1.4e18 and len(a["points"]) == 51
    assert np.linalg.norm(a["gt_v"], axis=1).max() > 0.5                   # a flying MAV, not the hovering synthetic platform
    P, V, R = a["gt_p"][0].copy(), a["gt_v"][0].copy(), a["gt_R"][0].copy()
    for k in range(1, 51):
        acc, gyr = a["imu"][k]
        for i in range(1, len(acc)):
            a0 = R @ (acc[i - 1] - a["ba"]) - replay.G
            R = R @ replay.deltaQ_R((0.5 * (gyr[i - 1] + gyr[i]) - a["bg"]) * 0.005)
            u, _, vt = np.linalg.svd(R); R = u @ vt
            am = 0.5 * (a0 + R @ (acc[i] - a["ba"]) - replay.G)
            P = P + 0.005 * V + 0.5 * 0.005 ** 2 * am; V = V + 0.005 * am
    assert np.linalg.norm(P - a["gt_p"][50]) < 0.01 and np.linalg.norm(V - a["gt_v"][50]) < 0.01
    assert np.linalg.norm(replay._rotvec(R.T @ a["gt_R"][50])) < 5e-4
    with np.testing.assert_raises(ValueError):
        replay.simulate_stream_euroc("V2_02_medium", 400)


def test_replay_on_a_euroc_trajectory_with_the_oracle_back_end():
    """5 s of V1_03_difficult (fast rotations): both marginalisation modes occur, ~60 line factors per window from the sequence's
    own map; the estimate stays within a few cm of the ground truth (the paper's Table II reports 6.8 cm on a whole sequence)."""
    stream = replay.simulate_stream_euroc("V1_03_difficult", 50, start_s=1.0, max_features=40, max_lines=6)
    out = replay.run(stream, OracleBackend(), num_iterations=8)
    flags = [l["flag"] for l in out["log"]]
    assert replay.MARGIN_OLD in flags and replay.MARGIN_SECOND_NEW in flags
    assert min(l["n_line"] for l in out["log"]) >= 30
    i, j = ate.associate(out["t"], stream["t"])
    assert len(i) == 40
    assert ate.ate_rmse(out["p"][i], stream["gt_p"][j]) < 0.08
    assert ate.ate_rmse(out["p"][i], stream["gt_p"][j], align=False) < 0.25


def test_lock_step_replay_keeps_the_prior_maps_of_the_sequences_apart():
    """two sequences with DIFFERENT prior line maps in one lock-step replay, association in the loop: each is matched against
    its own map (same result as replayed alone)."""
    streams = [replay.simulate_stream_euroc(s, 16, start_s=1.0, max_features=30, max_lines=5, associate=True) for s in ("V1_02_medium", "V2_01_easy")]
    assert streams[0]["map_lines"].shape != streams[1]["map_lines"].shape
    many = replay.run_many(streams, OracleBackend(), num_iterations=4)
    for st, m in zip(streams, many):
        one = replay.run(st, OracleBackend(), num_iterations=4)
        assert np.array_equal(one["p"], m["p"]) and [l["n_line"] for l in one["log"]] == [l["n_line"] for l in m["log"]]
        assert all(l["n_line"] > 0 for l in m["log"])


def test_native_estimator_fails_loudly_without_a_device():
    """include/tcv_estimator.h on a box without a GPU: the window fills (host logic), the optimisation reports TCV_ERR_NO_DEVICE --
    no CPU solver hides behind the native window management either."""
    import ctypes as C
    import tcv
    if tcv.lib().tcv_device_count() > 0:
        return
    st = replay.simulate_stream(5, 12, max_features=24)
    try:
        replay.run_many_native([st], num_iterations=2)
    except RuntimeError as e:
        assert "no CPU path" in str(e) or "NO_DEVICE" in str(e) or "no HIP device" in str(e)
    else:
        raise AssertionError("the native estimator ran without a device")
    # the host side alone: ten frames fill the window, the eleventh is ready for the solver, which then refuses
    L = tcv.lib()
    cfg = replay._EstimatorConfig()
    cfg.focal_length = 460.0; cfg.min_parallax = 10.0 / 460.0; cfg.init_depth = 5.0; cfg.imu_dt = 0.005; cfg.num_iterations = 2
    cfg.gravity[:] = [0.0, 0.0, 9.81]; cfg.ric[:] = [1, 0, 0, 0, 1, 0, 0, 0, 1]; cfg.K[:] = [460, 0, 376, 0, 460, 240, 0, 0, 1]; cfg.width, cfg.height = 752, 480
    h = C.c_void_p()
    L.tcv_estimator_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(replay._EstimatorConfig)]
    assert L.tcv_estimator_create(C.byref(h), C.byref(cfg)) == 0
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    L.tcv_estimator_begin_frame.argtypes = [C.c_void_p, C.c_int, dp, dp, C.c_int, ip, dp, C.c_int, ip, dp, dp, ip]
    L.tcv_estimators_optimize.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.tcv_estimator_destroy.argtypes = [C.c_void_p]; L.tcv_estimator_destroy.restype = None
    ready = []
    for k in range(11):
        imu = st["imu"][k]
        acc = None if imu is None else np.ascontiguousarray(imu[0]); gyr = None if imu is None else np.ascontiguousarray(imu[1])
        ids = np.ascontiguousarray(list(st["points"][k].keys()), dtype=np.int32); pv = np.ascontiguousarray(np.array(list(st["points"][k].values())))
        truth = np.ascontiguousarray(np.concatenate([st["gt_p"][k], st["gt_R"][k].reshape(9), st["gt_v"][k]]))
        r = C.c_int(-1)
        assert L.tcv_estimator_begin_frame(h, 0 if acc is None else len(acc) - 1, None if acc is None else acc.ctypes.data_as(dp), None if gyr is None else gyr.ctypes.data_as(dp),
                                           len(ids), ids.ctypes.data_as(ip), pv.ctypes.data_as(dp), 0, None, None, truth.ctypes.data_as(dp), C.byref(r)) == 0
        ready.append(r.value)
    assert ready == [0] * 10 + [1]
    arr = (C.c_void_p * 1)(h)
    assert L.tcv_estimators_optimize(arr, 1) == tcv.TCV_ERR_NO_DEVICE
    bad = C.c_int()
    assert L.tcv_estimator_begin_frame(h, 3, None, None, 0, None, None, 0, None, None, None, C.byref(bad)) == tcv.TCV_ERR_INVALID
    L.tcv_estimator_destroy(h)


def test_bench_cpu_leg_of_the_replay_mode_runs_the_same_streams():
    """bench.py --mode replay, CPU baseline leg: a worker process replays stream i with the oracle back end and times the back end's calls only
    (no device involved); the generator arguments are the GPU run's, so the frame count must fit the 36 s ground-truth excerpts"""
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import bench
    n, t = bench._cpu_replay_worker((1, replay.WINDOW_SIZE + 1 + 10 + 100, 60, 8, 3.0))      # the default replay line: warm-up 10 + 100 steps
    assert n >= 3 and 0.0 < t < 3.5
    # the default line's replay figure (10 + 60 frames) and a stream of the second lap over the sequences (start offset 3 s)
    n2, t2 = bench._cpu_replay_worker((6, replay.WINDOW_SIZE + 1 + 10 + 60, 60, 8, 1.0))
    assert n2 >= 1 and t2 > 0.0


def test_native_estimator_host_side_entry_points_without_a_device():
    """tcv_estimators_begin_frames (one call per lock-step frame) and the window tap's error path are host code: they run without a GPU"""
    import ctypes as C
    import tcv
    L = tcv.lib()
    st = replay.simulate_stream(5, 6, max_features=20)
    cfg = replay._EstimatorConfig()
    cfg.focal_length = 460.0; cfg.min_parallax = replay.MIN_PARALLAX; cfg.init_depth = replay.INIT_DEPTH
    cfg.acc_n = cfg.gyr_n = cfg.acc_w = cfg.gyr_w = 1e-3
    cfg.gravity[:] = [0.0, 0.0, 9.81]; cfg.imu_dt = 0.005; cfg.K[:] = [460.0, 0, 376.0, 0, 460.0, 240.0, 0, 0, 1.0]; cfg.width = 752; cfg.height = 480
    cfg.tic[:] = [0.0, 0.0, 0.0]; cfg.ric[:] = [1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0]; cfg.estimate_extrinsic = 1
    cfg.angle_th, cfg.overlap_th, cfg.dist_th = 0.17, 0.45, 50.0
    cfg.num_iterations = 8; cfg.fixed_iterations = 0; cfg.line_exact_jacobian = 0
    vp = C.c_void_p
    L.tcv_estimator_create.argtypes = [C.POINTER(vp), C.POINTER(replay._EstimatorConfig)]
    L.tcv_estimator_destroy.argtypes = [vp]; L.tcv_estimator_destroy.restype = None
    L.tcv_estimators_begin_frames.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(replay._FrameInput), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.tcv_estimator_set_window_tap.argtypes = [vp, C.c_int]
    hs = [vp(), vp()]
    for h in hs:
        tcv.check(L.tcv_estimator_create(C.byref(h), C.byref(cfg)))
    try:
        arr = (vp * 2)(*hs)
        rec = (replay._FrameInput * 2)()
        pts = np.ascontiguousarray(np.array([[0.1, 0.2, 1.0], [-0.1, 0.05, 1.0]])); ids = np.ascontiguousarray([7, 9], dtype=np.int32)
        for r in rec:
            r.n_imu = 0; r.n_points = 2; r.point_ids = ids.ctypes.data_as(C.POINTER(C.c_int)); r.points = pts.ctypes.data_as(C.POINTER(C.c_double)); r.n_lines = 0
        ready = (C.c_int * 2)(5, 5); rcs = (C.c_int * 2)(9, 9)
        tcv.check(L.tcv_estimators_begin_frames(arr, 2, rec, ready, rcs))
        assert list(ready) == [0, 0] and list(rcs) == [0, 0]                      # the window is not full after one frame
        assert L.tcv_estimators_begin_frames((vp * 2)(hs[0], hs[0]), 2, rec, ready, rcs) != 0      # the same estimator twice
        tcv.check(L.tcv_estimator_set_window_tap(hs[0], 1))
        snap = (C.c_char * 1024)()
        L.tcv_estimator_get_window_snapshot.argtypes = [vp, C.c_void_p]
        assert L.tcv_estimator_get_window_snapshot(hs[0], C.cast(snap, C.c_void_p)) != 0            # no window optimised yet
        assert b"no snapshot" in L.tcv_last_error()
    finally:
        for h in hs:
            L.tcv_estimator_destroy(h)
