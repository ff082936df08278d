"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/tcv.h declares,
the ceres::Problem-shaped host logic works without a device, and every compute entry point fails loudly
(TCV_ERR_NO_DEVICE) instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import synth
from util import golden_windows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tcv(built):
    import tcv
    return tcv


def test_library_exports_every_declared_symbol(tcv):
    hdr = open(os.path.join(ROOT, "include", "tcv.h")).read()
    declared = set(re.findall(r"\b(tcv_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 40
    L = tcv.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"libtcv_hip.so does not export {name}"
    assert declared == set(tcv.EXPORTS)
    assert b"gfx950" in L.tcv_version()
    # include/tcv_estimator.h: the native window management around the same C-ABI
    est = set(re.findall(r"\b(tcv_estimators?_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "tcv_estimator.h")).read()))
    assert len(est) == 17
    for name in sorted(est):
        assert hasattr(L, name), f"libtcv_hip.so does not export {name}"


def test_product_does_not_link_or_import_the_oracle(tcv):
    import subprocess
    out = subprocess.check_output(["ldd", tcv.LIB_PATH]).decode()
    assert "liborc" not in out
    src = "".join(open(os.path.join(ROOT, "tc-viml_amd", f)).read() for f in ("tcv.py", "synth.py", "replay.py", "ate.py", "build.py"))
    assert "import orc" not in src and "np_oracle" not in src and "from oracle" not in src


def test_window_graph_counts(tcv):
    b = synth.make_windows(5, 1)
    w = tcv.Window(synth.window_at(b, 0))
    L = tcv.lib()
    # estimator.cpp:1683-1846: 11 poses + 11 speed-biases + extrinsic + 50 inverse depths; 10 IMU + 200 point + 40 line blocks
    assert L.tcv_problem_num_parameter_blocks(w.h) == 11 + 11 + 1 + 50
    assert L.tcv_problem_num_residual_blocks(w.h) == 10 + 200 + 40
    assert L.tcv_problem_num_residuals(w.h) == 150 + 400 + 80
    st = w.plan_stats()
    # default = chain layout: the 11 speed-bias blocks are eliminated one by one, the pose system (72 + rhs row) is 5 tile rows,
    # and the whole window fits half a CU's LDS (two windows per CU)
    assert st["nc"] == 171 and st["nx"] == 183 and st["npp"] == 72 and st["nland"] == 50 and st["nt"] == 5
    assert st["n_iunit"] == 11 and st["lds_bytes"] == 80 * 1024
    L.tcv_set_solver_variant(1)                   # dense layout: one 171 (+ rhs) system = 11 tile rows, the whole LDS
    try:
        sd = tcv.Window(synth.window_at(b, 0)).plan_stats()
    finally:
        L.tcv_set_solver_variant(0)
    assert sd["nt"] == 11 and sd["lds_bytes"] <= 160 * 1024
    # algorithmic window data of cfg 2 (no prior): SURVEY.md 8(d) counts 39 728 B incl. indices; the data pool holds the doubles
    assert 4500 <= st["window_doubles"] <= 5000


def test_prior_and_constant_extrinsic_change_the_plan(tcv):
    pre, main, z = golden_windows()
    w = tcv.Window(main)
    st = w.plan_stats()
    # + the prior: J0 | r0 WITHOUT the leading rows that are exact zeros (the eigenvalues the marginalisation thresholded,
    # marginalization_factor.cpp:284-293), + x0
    J0g, r0g = z["marg_J0"], z["marg_r0"]
    k0 = 0
    while k0 < J0g.shape[0] - 1 and not J0g[k0].any() and r0g[k0] == 0.0:
        k0 += 1
    nfull, nr = J0g.shape[0], J0g.shape[0] - k0
    assert 10 < k0 < nfull - 10                  # about half of the 75 rows of the golden prior
    st0 = tcv.Window(pre).plan_stats()
    extra = st["window_doubles"] - st0["window_doubles"]
    assert nr * nfull + nr <= extra <= nr * nfull + nr + 200, (extra, nr, nfull)
    w2 = tcv.Window(main, estimate_extrinsic=False)
    st2 = w2.plan_stats()
    assert st2["nc"] == 165 and st2["npp"] == 66
    m, n, nb, xs = w.prior.dims()
    assert (m, n, nb) == (int(z["marg_m"]), int(z["marg_n"]), len(z["marg_sizes"]))
    d = w.prior.export()
    assert np.array_equal(d["J0"], z["marg_J0"]) and d["idx"] == [int(i) for i in z["marg_idx"]]


def test_large_landmark_count_is_chunked_not_rejected(tcv):
    b = synth.make_windows(9, 1, n_landmarks=200)
    w = tcv.Window(synth.window_at(b, 0))
    st = w.plan_stats()
    assert st["nland"] == 200 and st["n_vis_chunk"] >= 2


def test_error_paths(tcv):
    L = tcv.lib()
    p = C.c_void_p()
    assert L.tcv_problem_create(C.byref(p)) == 0
    x = np.zeros(7); x[6] = 1
    assert L.tcv_problem_add_parameter_block(p, tcv.dptr(x), 6, tcv.TCV_PARAM_POSE) == tcv.TCV_ERR_INVALID     # pose blocks are size 7
    assert L.tcv_problem_add_parameter_block(p, tcv.dptr(x), 7, tcv.TCV_PARAM_POSE) == 0
    assert L.tcv_problem_add_parameter_block(p, tcv.dptr(x), 9, tcv.TCV_PARAM_EUCLIDEAN) == tcv.TCV_ERR_INVALID  # re-add with another size
    y = np.zeros(3)
    assert L.tcv_problem_set_parameter_block_constant(p, tcv.dptr(y)) == tcv.TCV_ERR_INVALID                  # unknown block
    assert b"unknown block" in L.tcv_last_error()
    out = np.zeros(16, np.int32)
    # a lone pose block is not a window the LDS-resident solver can stage: reported, never a crash
    assert L.tcv_problem_plan_stats(p, tcv.iptr(out)) in (0, tcv.TCV_ERR_TOO_LARGE)
    L.tcv_problem_destroy(p)
    # NaN in the window data is reported, not propagated
    b = synth.make_windows(3, 1)
    w0 = synth.window_at(b, 0)
    w0["pose"] = w0["pose"].copy(); w0["pose"][2, 1] = np.nan
    w = tcv.Window(w0)
    assert L.tcv_problem_plan_stats(w.h, tcv.iptr(out)) == tcv.TCV_ERR_NUMERIC


def test_null_addresses_are_rejected_not_resolved_to_block_zero(tcv):
    """A NULL parameter-block address never names a block (ceres::Problem CHECKs on it): the open-addressing index marks empty slots with
    nullptr, so a NULL key used to match the first empty slot and come back as block 0 (round-5 advisor finding)."""
    L = tcv.lib()
    w = tcv.Window(synth.window_at(synth.make_windows(3, 1), 0))
    null = C.POINTER(C.c_double)()
    assert L.tcv_problem_set_parameter_block_constant(w.h, null) == tcv.TCV_ERR_INVALID
    assert b"unknown block" in L.tcv_last_error()
    assert L.tcv_problem_add_parameter_block(w.h, null, 7, tcv.TCV_PARAM_POSE) == tcv.TCV_ERR_INVALID
    nf = w.desc.n_frames
    PP = C.POINTER(C.c_double) * nf
    pose_rows = [tcv.dptr(w.pose[i]) for i in range(nf)]
    good = PP(*pose_rows)
    assert L.tcv_problem_set_frames(w.h, nf, good, None) == 0
    rows = list(pose_rows); rows[3] = null
    assert L.tcv_problem_set_frames(w.h, nf, PP(*rows), None) == tcv.TCV_ERR_INVALID and b"unknown pose block" in L.tcv_last_error()
    # a marginalisation factor over a NULL block
    n = 6
    pr = C.c_void_p()
    size = tcv.i32(np.array([7])); idx = tcv.i32(np.array([0]))
    assert L.tcv_prior_create(C.byref(pr), 0, n, 1, tcv.iptr(size), tcv.iptr(idx), tcv.dptr(np.zeros(7)), tcv.dptr(np.eye(n)), tcv.dptr(np.zeros(n))) == 0
    blocks = (C.POINTER(C.c_double) * 1)(null)
    assert L.tcv_problem_add_marginalization_factor(w.h, pr, blocks, 1) == tcv.TCV_ERR_INVALID
    L.tcv_prior_destroy(pr)


def test_window_description_is_validated_before_it_is_walked(tcv):
    """tcv_problem_from_window / tcv_batch_create: missing arrays and out-of-range frame / feature indices are reported as
    TCV_ERR_INVALID with a message, never dereferenced."""
    L = tcv.lib()
    w = tcv.Window(synth.window_at(synth.make_windows(3, 1), 0))
    h = C.c_void_p()

    def broken(**kw):
        d = tcv.WindowDesc()
        C.memmove(C.byref(d), C.byref(w.desc), C.sizeof(d))
        for k, v in kw.items():
            setattr(d, k, v)
        return L.tcv_problem_from_window(C.byref(d), C.byref(h))

    assert broken() == 0
    L.tcv_problem_destroy(h)
    assert broken(para_pose=None) == tcv.TCV_ERR_INVALID and b"missing array" in L.tcv_last_error()
    assert broken(proj_pts=None) == tcv.TCV_ERR_INVALID
    assert broken(imu=None) == tcv.TCV_ERR_INVALID
    assert broken(line_data=None) == tcv.TCV_ERR_INVALID
    assert broken(n_proj=-1) == tcv.TCV_ERR_INVALID
    bad = tcv.i32(np.full(w.desc.n_proj, 11)); assert broken(proj_frame_j=tcv.iptr(bad)) == tcv.TCV_ERR_INVALID and b"out of range" in L.tcv_last_error()
    bad = tcv.i32(np.full(w.desc.n_proj, w.desc.n_landmarks)); assert broken(proj_feature=tcv.iptr(bad)) == tcv.TCV_ERR_INVALID
    bad = tcv.i32(np.full(w.desc.n_line, -1)); assert broken(line_frame=tcv.iptr(bad)) == tcv.TCV_ERR_INVALID
    bad = tcv.i32(np.full(w.desc.n_imu, 12)); assert broken(imu_frame_i=tcv.iptr(bad)) == tcv.TCV_ERR_INVALID
    b = C.c_void_p()
    arr = (C.c_void_p * 2)(w.h, None)
    assert L.tcv_batch_create(C.byref(b), arr, None, None, None, 2) == tcv.TCV_ERR_INVALID and b"null problem" in L.tcv_last_error()


def test_time_offset_windows_in_both_layouts(tcv):
    """ESTIMATE_TD: para_Td rides behind the poses in tangent space (nc = 172) as the last column of the pose part (npp = 73): the chain
    layout keeps it with the poses (5 tile rows: 73 columns + rhs, Td's six-wide gather slot inside them), solver variant 1 lays the plan out
    for the dense kernel (11 tile rows); mixing ProjectionFactors and ProjectionTdFactors in one problem is refused."""
    L = tcv.lib()
    w = synth.with_time_offset(synth.window_at(synth.make_windows(3, 1), 0), 3)
    W = tcv.Window(w)
    st = W.plan_stats()
    assert st["nc"] == 172 and st["npp"] == 73 and st["nt"] == 5 and st["lds_bytes"] == 80 * 1024
    L.tcv_set_solver_variant(1)
    try:
        st = tcv.Window(w).plan_stats()
        assert st["nc"] == 172 and st["npp"] == 73 and st["nt"] == 11 and st["lds_bytes"] == 160 * 1024
    finally:
        L.tcv_set_solver_variant(0)
    # a prior that holds Td (n = 76, block kind 3) ties it to the first speed-bias block: Td becomes one more column of that chain
    # step's front, the plan stays a chain plan
    import np_oracle as NO
    po, _ = NO.marginalize_old(NO.Problem(w), NO.Problem(w).x0())
    assert po["n"] == 76 and tuple(po["blocks"][-1]) == ("td", 0)
    nxt = synth.with_time_offset(synth.window_at(synth.make_windows(4, 1), 0), 4)
    keep = dict(po, blocks=[tuple(bk) for bk in po["blocks"]])
    keep["x0"] = [np.array({"pose": nxt["pose"], "sb": nxt["speedbias"]}[nm][i], dtype=float).copy() if nm in ("pose", "sb")
                  else (np.array(nxt["ex_pose"], dtype=float).copy() if nm == "ex" else np.array([0.0])) for nm, i in keep["blocks"]]
    st2 = tcv.Window(dict(nxt, prior=keep)).plan_stats()
    assert st2["nt"] == 5 and st2["lds_bytes"] == 80 * 1024 and st2["window_doubles"] > st["window_doubles"] + 76 * 40      # (the prior without its zero rows)
    assert L.tcv_problem_num_parameter_blocks(W.h) == 11 * 2 + 1 + 1 + 50
    pts = np.array([0.1, 0.2, 1.0])
    assert L.tcv_problem_add_projection_factor(W.h, tcv.dptr(pts), tcv.dptr(pts), 306.0, 1.0, W.block_ptr("pose", 0), W.block_ptr("pose", 1),
                                               W.block_ptr("ex", 0), W.block_ptr("lam", 0)) == 0
    out = np.zeros(16, np.int32)
    assert L.tcv_problem_plan_stats(W.h, tcv.iptr(out)) == tcv.TCV_ERR_UNSUPPORTED and b"ProjectionTdFactors" in L.tcv_last_error()
    assert L.tcv_problem_set_rolling_shutter(W.h, 0.02, 0.0) == tcv.TCV_ERR_INVALID


def test_too_many_frames_is_rejected(tcv):
    # 13 frames -> camera tangent dim 201 > 175: the LDS-resident solver refuses instead of truncating
    L = tcv.lib()
    p = C.c_void_p(); L.tcv_problem_create(C.byref(p))
    poses = np.zeros((13, 7)); poses[:, 6] = 1; sbs = np.zeros((13, 9))
    for i in range(13):
        L.tcv_problem_add_parameter_block(p, tcv.dptr(poses[i]), 7, tcv.TCV_PARAM_POSE)
        L.tcv_problem_add_parameter_block(p, tcv.dptr(sbs[i]), 9, tcv.TCV_PARAM_EUCLIDEAN)
    out = np.zeros(16, np.int32)
    assert L.tcv_problem_plan_stats(p, tcv.iptr(out)) == tcv.TCV_ERR_TOO_LARGE
    L.tcv_problem_destroy(p)


def test_compute_entry_points_fail_loudly_without_a_device(tcv):
    L = tcv.lib()
    if L.tcv_device_count() > 0:
        pytest.skip("a HIP device is visible")
    b = synth.make_windows(0, 1)
    w = tcv.Window(synth.window_at(b, 0))
    with pytest.raises(tcv.TcvError) as e:
        tcv.Batch([w])
    assert e.value.status == tcv.TCV_ERR_NO_DEVICE
    with pytest.raises(tcv.TcvError) as e:
        tcv.pose_plus(np.zeros((1, 7)), np.zeros((1, 6)))
    assert e.value.status == tcv.TCV_ERR_NO_DEVICE
    s = tcv.SolverSummary()
    o = tcv.default_options()
    assert L.tcv_solve(C.byref(o), w.h, C.byref(s)) == tcv.TCV_ERR_NO_DEVICE
    assert np.array_equal(w.pose, synth.window_at(b, 0)["pose"])      # caller state untouched


def test_synthetic_generator_is_deterministic_and_batch_independent():
    a = synth.make_windows(40, 3)
    b = synth.make_windows(41, 1)
    w1 = synth.window_at(a, 1); w2 = synth.window_at(b, 0)
    for k in ("pose", "speedbias", "lam"):
        assert np.array_equal(w1[k], w2[k])
    assert np.array_equal(w1["imu"]["covariance"], w2["imu"]["covariance"])
    assert np.array_equal(w1["line"]["abc"], w2["line"]["abc"])
    assert len(w1["proj"]["frame_i"]) == 200 and len(w1["line"]["frame"]) == 40 and len(w1["imu"]["frame_i"]) == 10
    # consecutive tracks starting at the anchor (estimator.cpp:1745-1770), anchors < WINDOW_SIZE - 2 (:1740)
    assert np.all(w1["proj"]["frame_j"] > w1["proj"]["frame_i"]) and w1["proj"]["frame_i"].max() < 8


def test_only_tests_smoke_and_cpu_baseline_touch_the_oracle():
    """the oracle is test infrastructure: nothing under tc-viml_amd/ or tools/ imports it, bench.py only inside cpu_baseline()."""
    import glob
    for f in glob.glob(os.path.join(ROOT, "tc-viml_amd", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "*.py")):
        src = open(f).read()
        assert "import orc" not in src and "import np_oracle" not in src and "replay_oracle" not in src, f
    b = open(os.path.join(ROOT, "bench.py")).read()
    # the single import sits in the worker of the CPU baseline, behind the "CPU baseline" banner and before the PCIe / mode sections
    assert b.count("import orc") == 1 and b.index("# ---- CPU baseline") < b.index("import orc") < b.index("def cpu_baseline") < b.index("# ---- PCIe-inclusive")
    assert b.count("_cpu_worker") == 3      # its definition and the two calls inside cpu_baseline()
    for f in glob.glob(os.path.join(ROOT, "tc-viml_amd", "csrc", "*")):
        src = open(f).read()
        assert "tcv_oracle" not in src and "orc_" not in src and "liborc" not in src, f      # comments may mention the oracle, code may not use it


def test_chain_layout_eligibility_is_decided_on_the_host(tcv):
    """the packer's symbolic elimination (no GPU needed): speed-bias chains get the chain layout; a prior that ties two
    non-adjacent speed-bias blocks together (more than one later-eliminated neighbour) falls back to the dense layout; a broken
    IMU chain (sum_dt > 10, estimator.cpp:1726) is still a forest of chains; the chain steps cover every speed-bias block."""
    pre, main, z = golden_windows()
    st = tcv.Window(main).plan_stats()
    assert st["nt"] == 5 and st["n_iunit"] == 11 and st["n_imu_chunk"] == 1 and st["n_vis_chunk"] == 3
    # prior on sb 0 and sb 5 as well: chain broken -> dense
    p = main["prior"]
    n = p["n"] + 9
    J0 = np.zeros((n, n)); J0[:p["n"], :p["n"]] = p["J0"]; J0[p["n"]:, p["n"]:] = np.eye(9)
    p2 = dict(p, n=n, J0=J0, r0=np.concatenate([p["r0"], np.zeros(9)]), sizes=list(p["sizes"]) + [9], idx=list(p["idx"]) + [p["n"]],
              x0=list(p["x0"]) + [np.asarray(main["speedbias"])[5].copy()], blocks=list(p["blocks"]) + [("sb", 5)])
    st2 = tcv.Window(dict(main, prior=p2)).plan_stats()
    assert st2["nt"] == 11 and st2["lds_bytes"] > 80 * 1024
    # ... but sb 0 and sb 1 (neighbours) keep it
    p3 = dict(p2, blocks=list(p["blocks"]) + [("sb", 1)], x0=list(p["x0"]) + [np.asarray(main["speedbias"])[1].copy()])
    st3 = tcv.Window(dict(main, prior=p3)).plan_stats()
    assert st3["nt"] == 5 and st3["n_iunit"] == 11
    # one IMU factor dropped (sum_dt > 10): two chains
    w4 = dict(pre); im = dict(pre["imu"]); sd = np.array(im["sum_dt"], dtype=float).copy(); sd[4] = 11.0; im["sum_dt"] = sd; w4["imu"] = im
    st4 = tcv.Window(w4).plan_stats()
    assert st4["nt"] == 5 and st4["n_iunit"] == 11
    # constant extrinsic: 66 pose dims, still 11 chain steps
    st5 = tcv.Window(main, estimate_extrinsic=False).plan_stats()
    assert st5["npp"] == 66 and st5["nt"] == 5 and st5["n_iunit"] == 11


def test_header_is_valid_c99_and_cxx_and_the_example_links(tcv, tmp_path):
    """include/tcv.h is the drop-in boundary: it must compile as plain C and as C++, and a C++ caller (examples/estimator_shim.cpp,
    the call sequence of a retargeted estimator.cpp) must link against libtcv_hip.so and fail loudly without a device."""
    import subprocess
    c = os.path.join(tmp_path, "t.c")
    open(c, "w").write('#include "tcv.h"\n#include "tcv_estimator.h"\nint main(void){ tcv_solver_options o; tcv_solver_options_default(&o); return (int)sizeof(tcv_solver_summary) > 0 ? 0 : 1; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + inc, "-fsyntax-only", c])
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Werror", "-I" + inc, "-x", "c++", "-fsyntax-only", c])
    exe = os.path.join(tmp_path, "shim")
    libdir = os.path.dirname(tcv.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-I" + inc, os.path.join(ROOT, "examples", "estimator_shim.cpp"), "-L" + libdir, "-ltcv_hip",
                           "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "6 parameter blocks, 2 residual blocks, 17 residuals" in out.stdout
    if tcv.lib().tcv_device_count() == 0:
        assert "no HIP device visible" in out.stdout          # the compute entry points refuse to run instead of falling back
    # the Ceres-shaped class veneer (include/tcv_ceres_shim.hpp) and its example
    exe2 = os.path.join(tmp_path, "shim2")
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Werror", "-I" + inc, os.path.join(ROOT, "examples", "estimator_shim_classes.cpp"), "-L" + libdir,
                           "-ltcv_hip", "-Wl,-rpath," + libdir, "-o", exe2])
    out2 = subprocess.run([exe2], capture_output=True, text=True, timeout=120)
    assert out2.returncode == 0, out2.stdout + out2.stderr


def test_prior_create_rejects_malformed_layouts(tcv):
    """keep_block_idx / keep_block_size of a caller-supplied prior index fixed-size device buffers: sizes must be positive, every
    block must lie in [m, m + n), blocks must not overlap and their local sizes must sum to n (MarginalizationInfo::localSize,
    marginalization_factor.cpp:100-103: a size-7 block is 6 wide)."""
    pre, main, z = golden_windows()
    good = dict(main["prior"])
    P = tcv.Prior.from_dict(good)
    assert P.dims()[:2] == (good["m"], good["n"])

    def rc_of(**kw):
        p = dict(good, **kw)
        try:
            tcv.Prior.from_dict(p)
        except tcv.TcvError as e:
            return e.status
        return 0

    sizes, idx = list(good["sizes"]), list(good["idx"])
    assert rc_of(sizes=[0] + sizes[1:]) == tcv.TCV_ERR_INVALID                        # non-positive size
    assert rc_of(idx=[idx[0] - 1] + idx[1:]) == tcv.TCV_ERR_INVALID                    # below m
    assert rc_of(idx=idx[:-1] + [good["n"] - 3]) == tcv.TCV_ERR_INVALID                # runs past n
    assert rc_of(idx=[idx[1]] + idx[1:]) == tcv.TCV_ERR_INVALID                        # overlap
    assert rc_of(sizes=sizes[:-1], idx=idx[:-1], x0=good["x0"][:-1]) == tcv.TCV_ERR_INVALID   # local sizes do not sum to n
    big = dict(m=0, n=130, sizes=[130], idx=[0], x0=[np.zeros(130)], J0=np.zeros((130, 130)), r0=np.zeros(130))
    try:
        tcv.Prior.from_dict(big); rc = 0
    except tcv.TcvError as e:
        rc = e.status
    assert rc == tcv.TCV_ERR_TOO_LARGE


def test_solver_options_document_the_iteration_limit():
    hdr = open(os.path.join(ROOT, "include", "tcv.h")).read()
    assert "may exceed TCV_MAX_TRACE" in hdr and "workgroups_per_window" in hdr and "reserved, ignored" not in hdr


def test_no_static_lds_in_the_two_per_cu_kernels(tcv, tmp_path):
    """The chain-layout solve kernel and `marg_kernel<256>` run TWO workgroups per CU on exactly half a CU's LDS each (80 KiB, dynamic): a
    single static LDS variable -- e.g. the one `__syncthreads_or` brings for its work-group reduction -- makes it one workgroup per CU
    (round 4 measured that: marginalisation 0.85 -> 1.40 ms per 1024 windows).  The code objects' metadata must say
    `.group_segment_fixed_size: 0` for every kernel of those translation units."""
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    objdir = os.path.join(ROOT, "tc-viml_amd", "build", "libtcv_hip")
    seen = {}
    for obj in ("tcv_marg.hip.o", "tcv_solve_chain.o", "tcv_solve_chain_td.o"):
        path = os.path.join(objdir, obj)
        if not os.path.exists(path):
            pytest.skip("object files of the build are not there (library built elsewhere)")
        fat = os.path.join(tmp_path, obj + ".fat"); co = os.path.join(tmp_path, obj + ".co")
        subprocess.check_call([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, path])
        subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co])
        notes = subprocess.check_output([os.path.join(llvm, "llvm-readelf"), "--notes", co]).decode()
        lds = None
        for line in notes.splitlines():
            line = line.strip()
            if line.startswith(".group_segment_fixed_size:"):
                lds = int(line.split(":")[1])
            elif line.startswith(".name:") and lds is not None:
                seen[line.split(":", 1)[1].strip()] = lds
                lds = None
    big = {k: v for k, v in seen.items() if "marg_kernel" in k or "solve_kernel" in k}
    assert len(big) >= 3, seen
    assert all(v == 0 for v in big.values()), big


def test_device_memory_stats_without_a_device(tcv):
    """tcv_device_memory_stats is plain bookkeeping of the library's allocator: callable without a device, all zeros before any allocation"""
    live, cached, n = tcv.device_memory_stats()
    assert (live, cached, n) == (0, 0, 0)
