"""GPU parity against the pins that do not depend on the builder's reading of Ceres (tests/golden/pins.npz): the 50-digit mpmath
first step, finite-difference Jacobians from the residual formulas, the SciPy minimiser of the window cost."""
import numpy as np
import pytest

from util import fro, golden_windows, imu_pre, load

pytestmark = pytest.mark.gpu


def _solve(tcv, w, iters, fixed, **kw):
    W = tcv.Window(w, **kw)
    b = tcv.Batch([W])
    b.solve(tcv.default_options(iters, fixed, True, 256, True)); b.synchronize(); b.download_states()
    return W, b, b.summaries()[0]


@pytest.mark.parametrize("variant", [0, 1])
def test_first_step_matches_the_50_digit_solution(gpu, variant):
    """dx of iteration 1 -- landmark Schur, chain elimination of the speed-bias blocks (variant 0) or the dense 171-dim tiles
    (variant 1), tiled Cholesky, dogleg interpolation -- against the mpmath solution of the same scaled, mu-regularised system:
    north_star's 1e-6 on dx; measured ~1e-9."""
    pre, main, z = golden_windows()
    P = load("pins.npz")
    gpu.check(gpu.lib().tcv_set_solver_variant(variant))
    try:
        for w, p in ((pre, "pre_"), (main, "main_")):
            W, b, s = _solve(gpu, w, 1, True)
            assert b.plan_stats()["layout"] == ("chain", "dense")[variant]
            err = fro(b.first_step(0), P["mp_" + p + "delta"])
            print(p, "layout", variant, "first step vs 50-digit solution: %.2e" % err)
            assert err < 1e-7
            assert s.dogleg_case[1] == int(P["mp_" + p + "case"])
    finally:
        gpu.check(gpu.lib().tcv_set_solver_variant(0))


def test_device_jacobians_match_the_finite_difference_fixture(gpu):
    z = load("factors.npz"); P = load("pins.npz")
    params = np.concatenate([z["p1_pose_i"], z["p1_pose_j"], z["p1_ex"], z["p1_lam"][:, None]], 1)
    pts = np.concatenate([z["p1_pts_i"], z["p1_pts_j"]], 1)
    r, Js = gpu.eval_proj(pts, params, float(z["p1_sqrt_info"]))
    for k in np.nonzero(P["fd_p1_unit"])[0]:
        for b in range(3):
            assert np.abs(Js[b][k][:, :6] - P[f"fd_p1_J{b}"][k]).max() < 2e-5 * max(1.0, np.abs(Js[b][k]).max())
        assert np.abs(Js[3][k] - P["fd_p1_J3"][k]).max() < 2e-5 * max(1.0, np.abs(Js[3][k]).max())
    n = len(z["i1_sum_dt"])
    imu = {k: z["i1_" + k] for k in ("delta_p", "delta_q", "delta_v", "lin_ba", "lin_bg", "sum_dt", "jacobian", "covariance")}
    imu["frame_i"] = np.zeros(n, int)
    params = np.concatenate([z["i1_pose_i"], P["fd_i1_sb_i"], z["i1_pose_j"], z["i1_sb_j"]], 1)
    r, Js, _ = gpu.eval_imu(imu, params, z["i1_G"], sqrt_info=np.tile(np.eye(15), (n, 1, 1)))
    for k in np.nonzero(P["fd_i1_unit"])[0]:
        for b, wdt in enumerate((6, 9, 6, 9)):
            assert np.abs(Js[b][k][:, :wdt] - P[f"fd_i1_J{b}"][k]).max() < 1e-4 * max(1.0, np.abs(Js[b][k]).max())


def test_the_scipy_minimiser_is_a_fixed_point_and_a_lower_bound(gpu):
    """SciPy least_squares (trf, exact Jacobian of the robustified residuals) found the minimiser of the window cost without any
    dogleg / Ceres logic: tcv_solve started there must stop without moving the cost (<= 1e-9), started from the initial state it must
    end between that minimum and 1.01 x it (the restated dogleg stops on the function tolerance 0.85 % above the minimum, as the
    oracle does), and 100 forced iterations keep descending towards it."""
    pre, main, z = golden_windows()
    P = load("pins.npz")
    ln = main["line"]
    w = dict(main, line=dict(ln, frame=np.zeros(0, int), pts_start=np.zeros((0, 3)), pts_end=np.zeros((0, 3)), abc=np.zeros((0, 3))))
    cmin = float(P["sp_nolines_cost"])
    w2 = dict(w, pose=P["sp_nolines_pose"], speedbias=P["sp_nolines_sb"], ex_pose=P["sp_nolines_ex"], lam=P["sp_nolines_lam"])
    W, b, s = _solve(gpu, w2, 50, False)
    assert abs(s.initial_cost - cmin) < 1e-9 * cmin                      # the device evaluates the same cost at the same state
    assert s.termination in (1, 2, 3) and abs(s.final_cost - cmin) < 1e-9 * cmin
    W, b, s = _solve(gpu, w, 100, False)
    assert s.termination == 3 and cmin * (1 - 1e-9) <= s.final_cost < 1.01 * cmin
    W, b, s100 = _solve(gpu, w, 100, True)                               # max_num_iterations = 100 (sensor.yaml:86) honoured, no 63 clamp
    assert s100.num_iterations == 101 and cmin * (1 - 1e-9) <= s100.final_cost <= s.final_cost * (1 + 1e-9)
    # the opt-in exact line Jacobian makes the line factors part of a true least-squares problem: same fixed-point property
    we = dict(main, line=dict(ln, exact_jacobian=True))
    ce = float(P["sp_exact_cost"])
    we2 = dict(we, pose=P["sp_exact_pose"], speedbias=P["sp_exact_sb"], ex_pose=P["sp_exact_ex"], lam=P["sp_exact_lam"])
    W, b, s = _solve(gpu, we2, 50, False)
    assert abs(s.initial_cost - ce) < 1e-9 * ce and abs(s.final_cost - ce) < 1e-8 * ce


def test_marginalisation_schur_step_against_50_digits(gpu, monkeypatch):
    """tests/golden/marg_pin.npz (make_golden_marg_pin.py): A' = Arr - Arm Amm^-1 Amr and b' of the MARGIN_OLD factor set of the two golden
    windows at their initial states, accumulated and solved with mpmath at 50 digits from binary64 factor Jacobians.  Both routes of the
    kernel through Amm (Cholesky factor when the rank is proven; TCV_MARG_EIG_MM=1: Jacobi eigen-decomposition) against it.  For scale: the
    NumPy restatement with LAPACK's eigh -- the accuracy class of the reference's SelfAdjointEigenSolver on this graded matrix (eigenvalues
    1e1 ... 1e14: absolute accuracy eps |Amm|) -- is 7.4e-6 / 6.2e-7 away from the 50-digit result."""
    import ctypes as C
    z = load("marg_pin.npz")
    pre, main, _ = golden_windows()
    for w, p in ((pre, "pre_"), (main, "main_")):
        for mode in ("chol", "eig"):
            if mode == "eig":
                monkeypatch.setenv("TCV_MARG_EIG_MM", "1")
            else:
                monkeypatch.delenv("TCV_MARG_EIG_MM", raising=False)
            mw = gpu.margin_old_window(w)
            Wm = gpu.Window(mw)
            dr = gpu.margin_old_drops(Wm, mw)
            arr = (gpu._dp * len(dr))(*dr)
            h = C.c_void_p()
            gpu.check(gpu.lib().tcv_marginalize(Wm.h, arr, len(dr), C.byref(h)))
            P = gpu.Prior(h); As, bs = P.schur(); d = P.export()
            ea, eb = fro(As, z["mg_" + p + "A"]), fro(bs, z["mg_" + p + "b"])
            print("50-digit pin %s%s: A' %.2e b' %.2e, J0'J0 vs pin %.2e" % (p, mode, ea, eb, fro(d["J0"].T @ d["J0"], z["mg_" + p + "A"])))
            # measured: A' 2.2e-6 / 1.9e-6 (pre: Cholesky / eigen route), 2.1e-7 / 3.7e-7 (main) -- the floor set by the rounding of each side's own
            # Jacobians at these unsolved states; b' 6.9e-12 / 1.7e-8 (pre), 4.7e-13 / 3.5e-10 (main): the Cholesky route is four orders
            # closer to the exact Schur step than the eigen route, which in turn is three orders closer than LAPACK's eigh (1.6e-5 / 1.4e-6)
            assert ea < 1e-5 and eb < (1e-10 if mode == "chol" else 1e-6)
    monkeypatch.delenv("TCV_MARG_EIG_MM", raising=False)
