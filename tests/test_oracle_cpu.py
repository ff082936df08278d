"""CPU tests of the oracle (test infrastructure): the C restatement and the NumPy restatement against the
committed golden vectors, finite-difference checks where the reference Jacobian is a true derivative
(projection_factor.cpp:126-228 `check()` uses the same idea; never for the line factor, whose Jacobian is
"as written"), and the invariants the reference documents in commented-out asserts
(marginalization_factor.cpp:297-298)."""
import ctypes as C

import numpy as np
import pytest

import np_oracle as npo
import synth
import orc
from util import fro, golden_windows, imu_pre, load, rel


@pytest.fixture(scope="module")
def lib(built):
    return orc.lib()


def test_c_oracle_projection_matches_golden(lib):
    z = load("factors.npz")
    for k in range(len(z["p1_lam"])):
        r = np.zeros(2); J = [np.zeros((2, 7)), np.zeros((2, 7)), np.zeros((2, 7)), np.zeros((2, 1))]
        jp = (orc._dp * 4)(*[orc.dptr(j) for j in J])
        lib.orc_proj_evaluate(orc.dptr(z["p1_pose_i"][k].copy()), orc.dptr(z["p1_pose_j"][k].copy()), orc.dptr(z["p1_ex"][k].copy()),
                              float(z["p1_lam"][k]), orc.dptr(z["p1_pts_i"][k].copy()), orc.dptr(z["p1_pts_j"][k].copy()),
                              float(z["p1_sqrt_info"]), orc.dptr(r), jp)
        assert rel(r, z["p1_r"][k]) < 1e-11
        for b in range(4):
            assert rel(J[b], z[f"p1_J{b}"][k]) < 1e-11
            if b < 3:
                assert np.all(J[b][:, 6] == 0.0)      # 7th column zeroed (projection_factor.cpp:86,98,109)


def test_c_oracle_line_matches_golden(lib):
    z = load("factors.npz")
    K = np.ascontiguousarray(z["l1_K"]).reshape(9); R = np.ascontiguousarray(z["l1_Ric"]).reshape(9); T = np.ascontiguousarray(z["l1_Tic"])
    for k in range(len(z["l1_pose"])):
        r = np.zeros(2); J = np.zeros((2, 7))
        lc = np.concatenate([z["l1_start"][k], z["l1_end"][k], z["l1_abc"][k]])
        lib.orc_line_evaluate(orc.dptr(z["l1_pose"][k].copy()), orc.dptr(lc), orc.dptr(K), orc.dptr(R), orc.dptr(T), orc.dptr(r), orc.dptr(J))
        assert rel(r, z["l1_r"][k]) < 1e-11 and rel(J, z["l1_J"][k]) < 1e-11


def test_c_oracle_imu_matches_golden(lib):
    z = load("factors.npz")
    G = np.ascontiguousarray(z["i1_G"])
    for k in range(len(z["i1_sum_dt"])):
        im = {kk: z["i1_" + kk][k:k + 1] for kk in ("delta_p", "delta_q", "delta_v", "lin_ba", "lin_bg", "sum_dt", "jacobian", "covariance")}
        im["frame_i"] = np.zeros(1, int)
        c = orc.pack_imu_constants(im)[0].copy()
        S = np.ascontiguousarray(z["i1_sqrt_info"][k]).reshape(225)
        r = np.zeros(15); J = [np.zeros((15, 7)), np.zeros((15, 9)), np.zeros((15, 7)), np.zeros((15, 9))]
        jp = (orc._dp * 4)(*[orc.dptr(j) for j in J])
        lib.orc_imu_evaluate(orc.dptr(z["i1_pose_i"][k].copy()), orc.dptr(z["i1_sb_i"][k].copy()), orc.dptr(z["i1_pose_j"][k].copy()),
                             orc.dptr(z["i1_sb_j"][k].copy()), orc.dptr(c), orc.dptr(G), orc.dptr(S), orc.dptr(r), jp)
        assert rel(r, z["i1_r"][k]) < 1e-10
        for b in range(4):
            assert rel(J[b], z[f"i1_J{b}"][k]) < 1e-10
        # sqrt_info' sqrt_info == cov^-1 (imu_factor.h:64), C vs NumPy algorithms agree to ~cond * eps
        S_c = orc.imu_sqrt_info(z["i1_covariance"][k])
        assert rel(S_c.T @ S_c, np.linalg.inv(z["i1_covariance"][k])) < 1e-6
        assert rel(S_c, z["i1_sqrt_info"][k]) < 1e-5


def test_c_oracle_plus_and_corrector_match_golden(lib):
    z = load("factors.npz")
    for k in range(len(z["s2_x"])):
        out = np.zeros(7)
        lib.orc_pose_plus(orc.dptr(z["s2_x"][k].copy()), orc.dptr(z["s2_delta"][k].copy()), orc.dptr(out))
        assert rel(out, z["s2_out"][k]) < 1e-14
        assert abs(np.linalg.norm(out[3:]) - 1.0) < 1e-14     # Plus normalises (pose_local_parameterization.cpp:16)
    assert np.array_equal(z["s2_out"][0], z["s2_x"][0] / np.r_[1, 1, 1, [np.linalg.norm(z["s2_x"][0][3:])] * 4]) or rel(z["s2_out"][0], z["s2_x"][0]) < 1e-15
    for k in range(len(z["c1_r"])):
        r = z["c1_r"][k].copy(); J = z["c1_J"][k].copy()
        jp = (orc._dp * 1)(orc.dptr(J)); cols = (C.c_int * 1)(7)
        cost = lib.orc_loss_correct(2, orc.dptr(r), 1, jp, cols, C.c_double(1.0))
        assert rel(r, z["c1_r_out"][k]) < 1e-14 or np.all(z["c1_r_out"][k] == 0)
        assert rel(J, z["c1_J_out"][k]) < 1e-14
        assert abs(cost - z["c1_cost"][k]) <= 1e-14 * max(1.0, abs(z["c1_cost"][k]))
        s = float(z["c1_r"][k] @ z["c1_r"][k])
        assert abs(z["c1_cost"][k] - 0.5 * np.log1p(s)) < 1e-12 * max(1.0, s)     # Cauchy(a=1): rho0 = log(1+s)


def test_projection_jacobian_is_a_true_derivative():
    """central differences with Plus as perturbation, eps 1e-6 like ProjectionFactor::check (projection_factor.cpp:179)."""
    z = load("factors.npz")
    for k in (1, 2, 3, 4, 6, 7, 8, 9):      # k % 5 == 0 cases carry deliberately non-unit quaternions (Plus re-normalises)
        a, b, e, l = z["p1_pose_i"][k], z["p1_pose_j"][k], z["p1_ex"][k], float(z["p1_lam"][k])
        f = lambda a_, b_, e_, l_: npo.proj_evaluate(a_, b_, e_, l_, z["p1_pts_i"][k], z["p1_pts_j"][k], float(z["p1_sqrt_info"]), False)[0]
        h = 1e-6
        for blk, J in enumerate([z["p1_J0"][k], z["p1_J1"][k], z["p1_J2"][k]]):
            for c in range(6):
                d = np.zeros(6); d[c] = h
                xs = [a, b, e]
                xp = list(xs); xm = list(xs)
                xp[blk] = npo.pose_plus(xs[blk], d); xm[blk] = npo.pose_plus(xs[blk], -d)
                num = (f(xp[0], xp[1], xp[2], l) - f(xm[0], xm[1], xm[2], l)) / (2 * h)
                assert np.abs(num - J[:, c]).max() < 2e-5 * max(1.0, np.abs(J[:, c]).max())
        num = (f(a, b, e, l + h) - f(a, b, e, l - h)) / (2 * h)
        assert np.abs(num - z["p1_J3"][k][:, 0]).max() < 2e-5 * max(1.0, np.abs(z["p1_J3"][k]).max())


def test_imu_jacobian_is_a_true_derivative_at_the_linearisation_point():
    z = load("factors.npz")
    G = z["i1_G"]
    for k in range(0, 4):
        pre = imu_pre(z, k)
        S = np.eye(15)          # un-whitened: keeps the finite differences well scaled
        a, sa, b, sb = z["i1_pose_i"][k], z["i1_sb_i"][k].copy(), z["i1_pose_j"][k], z["i1_sb_j"][k]
        sa[3:] = np.concatenate([pre["lin_ba"], pre["lin_bg"]])      # first-order terms are exact only at the linearisation point
        r0, Js = npo.imu_evaluate(a, sa, b, sb, pre, G, sqrt_info=S)
        f = lambda a_, sa_, b_, sb_: npo.imu_evaluate(a_, sa_, b_, sb_, pre, G, sqrt_info=S, want_jac=False)[0]
        h = 1e-6
        for c in range(6):
            d = np.zeros(6); d[c] = h
            num = (f(npo.pose_plus(a, d), sa, b, sb) - f(npo.pose_plus(a, -d), sa, b, sb)) / (2 * h)
            assert np.abs(num - Js[0][:, c]).max() < 1e-4 * max(1.0, np.abs(Js[0][:, c]).max())
            num = (f(a, sa, npo.pose_plus(b, d), sb) - f(a, sa, npo.pose_plus(b, -d), sb)) / (2 * h)
            assert np.abs(num - Js[2][:, c]).max() < 1e-4 * max(1.0, np.abs(Js[2][:, c]).max())
        for c in range(9):
            d = np.zeros(9); d[c] = h
            num = (f(a, sa + d, b, sb) - f(a, sa - d, b, sb)) / (2 * h)
            assert np.abs(num - Js[1][:, c]).max() < 1e-4 * max(1.0, np.abs(Js[1][:, c]).max())
            num = (f(a, sa, b, sb + d) - f(a, sa, b, sb - d)) / (2 * h)
            assert np.abs(num - Js[3][:, c]).max() < 1e-4 * max(1.0, np.abs(Js[3][:, c]).max())


def test_c_oracle_solve_matches_golden_trace(lib):
    pre, main, z = golden_windows()
    for w, p in ((pre, "pre_"), (main, "main_")):
        O = orc.Window(w)
        s = O.solve(8, True)
        cost = np.array([s.cost[i] for i in range(s.num_iterations)])
        assert s.num_iterations == len(z[p + "cost"])
        assert rel(cost, z[p + "cost"]) < 1e-6                      # north_star tolerance on costs
        st = O.states()
        assert rel(st["pose"], z[p + "final_pose"]) < 1e-6 and rel(st["sb"], z[p + "final_sb"]) < 1e-6
        assert rel(st["lam"], z[p + "final_lam"]) < 1e-6 and rel(st["ex"], z[p + "final_ex"]) < 1e-6
        d1 = np.array(s.first_delta[:s.n_local])
        assert fro(d1, z[p + "first_delta"]) < 1e-6                 # dx of the scaled + regularised system
        assert [s.dogleg_case[i] for i in range(1, s.num_iterations)] == [int(c) for c in z[p + "case"][1:]]
        mc = np.array([s.model_cost_change[i] for i in range(1, s.num_iterations)])
        assert rel(mc, z[p + "model_cost_change"][1:]) < 1e-6


def test_c_oracle_convergence_mode_matches_golden(lib):
    pre, main, z = golden_windows()
    O = orc.Window(main)
    s = O.solve(50, False)
    assert s.num_iterations == int(z["conv_num_iterations"])
    assert s.termination == int(z["conv_termination"])
    assert abs(s.final_cost - float(z["conv_final_cost"])) < 1e-6 * float(z["conv_final_cost"])


def test_c_oracle_marginalisation_matches_golden_and_invariants(lib):
    pre, main, z = golden_windows()
    O = orc.Window(pre)
    O.solve(8, True)
    prior, dbg = O.marginalize_old()
    assert prior["m"] == int(z["marg_m"]) and prior["n"] == int(z["marg_n"])
    assert prior["sizes"] == [int(s) for s in z["marg_sizes"]] and prior["idx"] == [int(i) for i in z["marg_idx"]]
    # A' is a small difference of ~1e14 terms: ~1e-7 relative is the FP64 reproducibility floor
    assert fro(dbg["A_schur"], z["marg_A_schur"]) < 1e-5
    assert fro(dbg["b_schur"], z["marg_b_schur"]) < 1e-6
    JtJ = prior["J0"].T @ prior["J0"]; gJ = z["marg_J0"].T @ z["marg_J0"]
    assert fro(JtJ, gJ) < 1e-5
    # reference invariants (commented asserts marginalization_factor.cpp:297-298), up to the eps-thresholded null space
    assert fro(JtJ, dbg["A_schur"]) < 1e-5
    assert fro(prior["J0"].T @ prior["r0"], dbg["b_schur"]) < 1e-3
    assert rel(np.concatenate([np.atleast_1d(v) for v in prior["x0"]]), z["marg_x0"]) < 1e-6


def test_prior_residual_known_answer(lib):
    pre, main, z = golden_windows()
    O = orc.Window(main)
    r = np.zeros(int(z["marg_n"]))
    lib.orc_prior_residual(C.byref(O.c), orc.dptr(r))
    assert rel(r, z["m0_r"]) < 1e-9


def test_eig_sym_reconstructs(lib):
    rng = np.random.default_rng(3)
    for n in (1, 2, 7, 40):
        A = rng.normal(size=(n, n)); A = A @ A.T + 1e-3 * np.eye(n)
        a = A.copy().reshape(-1); ev = np.zeros(n); V = np.zeros(n * n)
        lib.orc_eig_sym(n, orc.dptr(a), orc.dptr(ev), orc.dptr(V))
        Vm = V.reshape(n, n).T        # column-major
        assert np.all(np.diff(ev) >= 0)
        assert rel(Vm @ np.diag(ev) @ Vm.T, A) < 1e-12


def test_margin_second_new_oracle_invariants(lib):
    """MARGIN_SECOND_NEW (estimator.cpp:2047-2113): prior-only marginalisation of para_Pose[WINDOW_SIZE-1]; the reference's
    documented invariants J0'J0 = A', J0'r0 = b' (marginalization_factor.cpp:297-298) and the Schur identity hold."""
    import np_oracle as NO
    import orc
    from util import fro, golden_windows
    pre, main, z = golden_windows()
    Oc = orc.Window(main); Oc.solve(8, True); st = Oc.states()
    w2 = dict(main, pose=st["pose"], speedbias=st["sb"], ex_pose=st["ex"], lam=st["lam"])
    prob = NO.Problem(w2)
    po, dbg = NO.marginalize_second_new(prob, prob.x0())
    assert (po["m"], po["n"]) == (6, main["prior"]["n"] - 6)
    assert ("pose", main["pose"].shape[0] - 2) not in [tuple(b) for b in po["blocks"]]
    assert fro(po["J0"].T @ po["J0"], dbg["A_schur"]) < 1e-9 and fro(po["J0"].T @ po["r0"], dbg["b_schur"]) < 1e-9
    # before the Schur complement the system is the old prior's own J0'J0 / J0'r at the current states
    r, Js, _ = prob.eval_factor(prob.factors()[0], prob.x0(), True)
    assert abs(np.trace(dbg["A"]) - sum(np.sum(J[:, :6 if J.shape[1] == 7 else J.shape[1]] ** 2) for J in Js)) < 1e-6 * np.trace(dbg["A"])
    # a window without a prior on pose WINDOW_SIZE-1 marginalises nothing (:2049-2050)
    assert NO.marginalize_second_new(NO.Problem(pre), NO.Problem(pre).x0()) == (None, None)


def test_c_oracle_projection_td_matches_golden_and_is_a_true_derivative(lib):
    """T1 ProjectionTdFactor (projection_td_factor.cpp:34-140): C restatement vs the NumPy golden vectors, the td column by
    central differences, and the degenerate case (zero feature velocity -> ProjectionFactor, zero td column)."""
    import ctypes as C
    import np_oracle as NO
    from util import load, rel
    z = load("proj_td.npz")
    n = z["pts"].shape[0]
    names = ["J_pose_i", "J_pose_j", "J_ex", "J_lam", "J_td"]
    for k in range(n):
        r = np.zeros(2); Js = [np.zeros(14), np.zeros(14), np.zeros(14), np.zeros(2), np.zeros(2)]
        arr = (C.POINTER(C.c_double) * 5)(*[J.ctypes.data_as(C.POINTER(C.c_double)) for J in Js])
        p = np.ascontiguousarray(z["params"][k]); pts = np.ascontiguousarray(z["pts"][k]); aux = np.ascontiguousarray(z["aux"][k])
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        lib.orc_proj_td_evaluate(dp(p[0:7].copy()), dp(p[7:14].copy()), dp(p[14:21].copy()), float(p[21]), float(p[22]), dp(pts[:3].copy()),
                                 dp(pts[3:].copy()), dp(aux), float(z["sqrt_info"]), float(z["TR"][k]), float(z["ROW"]), dp(r), arr)
        assert rel(r, z["res"][k]) < 1e-10
        for J, nm in zip(Js, names):
            assert rel(J.reshape(z[nm][k].shape), z[nm][k]) < 1e-9, (k, nm)
        # d r / d td by central differences
        f = lambda td: NO.proj_td_evaluate(p[0:7], p[7:14], p[14:21], p[21], td, pts[:3], pts[3:], aux[0:2], aux[2:4], aux[4], aux[5], aux[6],
                                           aux[7], float(z["sqrt_info"]), float(z["TR"][k]), float(z["ROW"]), False)[0]
        fd = (f(p[22] + 1e-6) - f(p[22] - 1e-6)) / 2e-6
        assert np.abs(fd - z["J_td"][k][:, 0]).max() < 1e-5 * max(1.0, np.abs(fd).max())
        if k >= 36:
            r0, J0 = NO.proj_evaluate(p[0:7], p[7:14], p[14:21], p[21], pts[:3], pts[3:], float(z["sqrt_info"]))
            assert np.array_equal(r0, z["res"][k]) and np.abs(z["J_td"][k]).max() == 0.0


def test_time_offset_window_in_the_numpy_oracle():
    """ESTIMATE_TD (estimator.cpp:1703-1707, :1757-1763): para_Td[0] is one more camera-side block, every point factor a
    ProjectionTdFactor.  The Td column of the stacked Jacobian is a true derivative (finite differences), the solve moves Td
    towards the offset the observations were generated with, and MARGIN_OLD keeps Td (prior n = 76, last block of size 1)."""
    w = synth.with_time_offset(synth.window_at(synth.make_windows(31, 1), 0), 31, td_true=0.004, TR=0.02)
    P = npo.Problem(w)
    assert P.nc == 172 and P.loff[("td", 0)] == 171
    x = P.x0()
    J, r, c = P.linearize(x)
    h = 1e-7
    xp = dict(x, td=x["td"] + h); xm = dict(x, td=x["td"] - h)
    # the robust corrector rescales J and r, so the finite difference is taken on the un-robustified residuals
    w0 = dict(w, proj=dict(w["proj"], loss_a=0.0)); P0 = npo.Problem(w0)
    J0, r0, _ = P0.linearize(x)
    fd = (P0.linearize(xp, want_jac=False)[1] - P0.linearize(xm, want_jac=False)[1]) / (2 * h)
    assert np.abs(J0[:, 171] - fd).max() < 1e-5 * max(1.0, np.abs(fd).max())
    x8, so = npo.solve(P, 8, True)
    assert so["final_cost"] < 1e-3 * so["initial_cost"]
    assert abs(x8["td"][0] - 0.004) < abs(x["td"][0] - 0.004)
    po, dbg = npo.marginalize_old(P, x8)
    assert po["n"] == 76 and po["sizes"][-1] == 1 and tuple(po["blocks"][-1]) == ("td", 0)
    assert np.abs(po["J0"].T @ po["J0"] - dbg["A_schur"]).max() < 1e-6 * np.abs(dbg["A_schur"]).max()


def test_exact_line_jacobian_is_the_derivative_of_the_reference_residual():
    """the opt-in extension tcv_problem_set_line_jacobian(p, 1): same residual as LineProjectionFactor, Jacobian = its derivative under
    PoseLocalParameterization::Plus (finite differences); the reference's own Jacobian (default) is not."""
    z = load("factors.npz")
    K, Ric, Tic = z["l1_K"], z["l1_Ric"], z["l1_Tic"]
    worst_exact, worst_ref = 0.0, 0.0
    for k in range(8):
        pose = z["l1_pose"][k]; ps, pe, abc = z["l1_start"][k], z["l1_end"][k], z["l1_abc"][k]
        r0, Je = npo.line_evaluate(pose, ps, pe, abc, K, Ric, Tic, True, exact=True)
        r1, Jr = npo.line_evaluate(pose, ps, pe, abc, K, Ric, Tic, True)
        assert np.array_equal(r0, r1)
        fd = np.zeros((2, 6)); h = 1e-6
        for c in range(6):
            d = np.zeros(6); d[c] = h
            fd[:, c] = (npo.line_evaluate(npo.pose_plus(pose, d), ps, pe, abc, K, Ric, Tic, False)[0] -
                        npo.line_evaluate(npo.pose_plus(pose, -d), ps, pe, abc, K, Ric, Tic, False)[0]) / (2 * h)
        worst_exact = max(worst_exact, np.abs(Je[0][:, :6] - fd).max() / np.abs(fd).max())
        worst_ref = max(worst_ref, np.abs(Jr[0][:, :6] - fd).max() / np.abs(fd).max())
    assert worst_exact < 1e-5 and worst_ref > 0.5


def test_relocalisation_pose_and_factors_in_the_numpy_oracle():
    """np_oracle.Problem with win["relo"] (estimator.cpp:1854-1886): relo_Pose is one more pose block, every matched landmark one more
    ProjectionFactor on (para_Pose[start], relo_Pose, para_Ex_Pose[0], para_Feature[idx]).  The relo columns of the window Jacobian agree with
    central differences under PoseLocalParameterization::Plus, and the solve lowers the cost and moves the relocalisation pose."""
    import np_oracle as NO
    import synth
    from relo_util import add_relocalisation
    w = add_relocalisation(dict(synth.window_at(synth.make_windows(9300, 1, with_lines=False), 0), prior=None), f=6, seed=2)
    prob = NO.Problem(w)
    assert prob.has_relo and prob.nc == 12 * 6 + 6 + 11 * 9 and len(w["relo"]["frame_i"]) >= 8
    n_relo = sum(1 for f in prob.factors() if f[0] == "proj_relo")
    assert n_relo == len(w["relo"]["frame_i"])
    x = prob.x0()
    lo = prob.loff[("relo", 0)]
    facs = [f for f in prob.factors() if f[0] == "proj_relo"]
    h = 1e-6
    for fac in facs[:4]:
        r, Js, _ = prob.eval_factor(fac, x, True)
        # undo nothing: compare the loss-corrected Jacobian of the relo block with differences of the loss-corrected residual is not meaningful
        # (the corrector is not a derivative), so difference the RAW residual
        kind, k, blks = fac
        xs = [prob.get(x, nm, i) for (nm, i) in blks]
        rl, pr = w["relo"], w["proj"]
        r0, J0 = NO.proj_evaluate(xs[0], xs[1], xs[2], float(xs[3][0]), rl["pts_i"][k], rl["pts_j"][k], pr["sqrt_info"], True)
        for c in range(6):
            d = np.zeros(6); d[c] = h
            rp, _ = NO.proj_evaluate(xs[0], NO.pose_plus(xs[1], d), xs[2], float(xs[3][0]), rl["pts_i"][k], rl["pts_j"][k], pr["sqrt_info"], False)
            rm, _ = NO.proj_evaluate(xs[0], NO.pose_plus(xs[1], -d), xs[2], float(xs[3][0]), rl["pts_i"][k], rl["pts_j"][k], pr["sqrt_info"], False)
            assert np.abs((rp - rm) / (2 * h) - J0[1][:, c]).max() < 2e-4 * max(1.0, np.abs(J0[1]).max())
    xs, so = NO.solve(prob, 8, True)
    assert so["final_cost"] < so["initial_cost"] and np.abs(xs["relo"] - w["relo"]["pose"]).max() > 1e-4
    J, r, cost = prob.linearize(x)
    assert np.abs(J[:, lo:lo + 6]).max() > 0 and J.shape[1] == prob.nlocal
