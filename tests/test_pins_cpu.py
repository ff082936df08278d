"""The oracles against the pins that do not depend on the builder's reading of Ceres (tests/golden/pins.npz, written by
tests/golden/make_golden_pins.py): the 50-digit mpmath first step, the SciPy minimiser of the window cost, the finite-difference
Jacobians from the residual formulas."""
import numpy as np
import pytest

import np_oracle as npo
import orc
from util import fro, golden_windows, imu_pre, load


@pytest.fixture(scope="module")
def lib(built):
    return orc.lib()


def _no_lines(win):
    ln = win["line"]
    return dict(win, line=dict(ln, frame=np.zeros(0, int), pts_start=np.zeros((0, 3)), pts_end=np.zeros((0, 3)), abc=np.zeros((0, 3))))


def test_first_step_of_both_oracles_matches_the_50_digit_solution(lib):
    """dx of the Jacobi-scaled, mu-regularised system + dogleg at radius 1e4, solved by 50-digit LU from the same J and r:
    north_star's 1e-6 with three digits of margin (1e-9)."""
    pre, main, z = golden_windows()
    P = load("pins.npz")
    for w, p in ((pre, "pre_"), (main, "main_")):
        s = orc.Window(w).solve(1, True)
        d = np.array(s.first_delta[:s.n_local])
        assert fro(d, P["mp_" + p + "delta"]) < 1e-7
        assert s.dogleg_case[1] == int(P["mp_" + p + "case"])
        assert fro(z[p + "first_delta"], P["mp_" + p + "delta"]) < 1e-7           # the NumPy oracle's (committed fixture)


def test_analytic_jacobians_match_the_finite_difference_fixture():
    """P1 and I1 Jacobians of the NumPy restatement against central differences of the residual formulas (the fixture),
    the check ProjectionFactor::check does (projection_factor.cpp:126-228)."""
    z = load("factors.npz"); P = load("pins.npz")
    for k in np.nonzero(P["fd_p1_unit"])[0]:
        for b in range(3):
            assert np.abs(z[f"p1_J{b}"][k][:, :6] - P[f"fd_p1_J{b}"][k]).max() < 2e-5 * max(1.0, np.abs(z[f"p1_J{b}"][k]).max())
        assert np.abs(z["p1_J3"][k] - P["fd_p1_J3"][k]).max() < 2e-5 * max(1.0, np.abs(z["p1_J3"][k]).max())
    G = z["i1_G"]
    for k in np.nonzero(P["fd_i1_unit"])[0][:8]:
        pre = imu_pre(z, k)
        r, Js = npo.imu_evaluate(z["i1_pose_i"][k], P["fd_i1_sb_i"][k], z["i1_pose_j"][k], z["i1_sb_j"][k], pre, G, sqrt_info=np.eye(15))
        for b, wdt in enumerate((6, 9, 6, 9)):
            assert np.abs(Js[b][:, :wdt] - P[f"fd_i1_J{b}"][k]).max() < 1e-4 * max(1.0, np.abs(Js[b]).max())


def test_the_scipy_minimiser_is_a_fixed_point_and_a_lower_bound(lib):
    """SciPy's trf minimiser of 0.5 sum rho(|r|^2) (no dogleg, no Ceres reading): started there, the restated Ceres loop must stop
    without moving the cost; started from the window's initial state it must never get below it.  (The restated dogleg crawls along
    the ill-conditioned valley -- pure Gauss-Newton steps with rho ~ 1 -- and stops on the function tolerance 0.85 % above the
    minimum)"""
    pre, main, z = golden_windows()
    P = load("pins.npz")
    w = _no_lines(main)
    cmin = float(P["sp_nolines_cost"])
    w2 = dict(w, pose=P["sp_nolines_pose"], speedbias=P["sp_nolines_sb"], ex_pose=P["sp_nolines_ex"], lam=P["sp_nolines_lam"])
    s = orc.Window(w2).solve(50, False)
    assert abs(s.initial_cost - cmin) < 1e-10 * cmin                     # same cost function
    assert s.termination in (1, 2, 3) and abs(s.final_cost - cmin) < 1e-9 * cmin
    s = orc.Window(w).solve(100, False)
    assert s.termination == 3 and cmin * (1 - 1e-9) <= s.final_cost < 1.01 * cmin
    we = dict(main, line=dict(main["line"], exact_jacobian=True))
    ce = float(P["sp_exact_cost"])
    we2 = dict(we, pose=P["sp_exact_pose"], speedbias=P["sp_exact_sb"], ex_pose=P["sp_exact_ex"], lam=P["sp_exact_lam"])
    prob = npo.Problem(we2)
    assert abs(prob.linearize(prob.x0(), want_jac=False)[2] - ce) < 1e-10 * ce


def test_marginalisation_schur_step_of_both_oracles_against_50_digits(lib):
    """tests/golden/marg_pin.npz: A' and b' of the MARGIN_OLD factor set of the golden windows at their initial states, Schur step at 50
    digits.  The C oracle (cyclic Jacobi: relative accuracy on the graded Amm) lands at the floor set by FP64 Jacobians; the NumPy
    oracle (LAPACK eigh, the accuracy class of the reference's SelfAdjointEigenSolver: absolute accuracy eps |Amm|) an order or more away."""
    z = load("marg_pin.npz")
    pre, main, _ = golden_windows()
    for w, p in ((pre, "pre_"), (main, "main_")):
        po, dbg = orc.Window(w).marginalize_old()
        ea, eb = fro(dbg["A_schur"], z["mg_" + p + "A"]), fro(dbg["b_schur"], z["mg_" + p + "b"])
        prob = npo.Problem(w)
        _, dn = npo.marginalize_old(prob, prob.x0())
        na, nb = fro(dn["A_schur"], z["mg_" + p + "A"]), fro(dn["b_schur"], z["mg_" + p + "b"])
        print("%s C oracle A' %.2e b' %.2e | NumPy oracle A' %.2e b' %.2e" % (p, ea, eb, na, nb))
        assert ea < 1e-5 and eb < 1e-9          # measured 2.5e-6 / 2.6e-7 and 3.6e-11 / 1.9e-12
        assert na < 5e-5 and nb < 1e-4          # measured 7.4e-6 / 6.2e-7 and 1.6e-5 / 1.4e-6
