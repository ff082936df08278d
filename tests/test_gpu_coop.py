"""GPU tests of the cooperative small-batch mode of the fused solver (include/tcv.h tcv_set_cooperative; tc-viml_amd/csrc/tcv_solve.hip
linearize_coop / coop_helper): a window on 1 + H workgroups.  The mode changes WHERE the point / line factors are evaluated and the
landmarks eliminated, not a single addition: the same batch solved with one workgroup per window (workgroups_per_window = 1) must
give the same bits; against the oracle the usual gates hold (every other small-batch GPU test runs in this mode by default)."""
import ctypes as C

import numpy as np
import pytest

import synth
from util import golden_windows, rel

pytestmark = pytest.mark.gpu


def _solve(gpu, wins, iters, fixed, wg, marg=False):
    W = [gpu.Window(w) for w in wins]
    if marg:
        MW = [gpu.margin_old_window(w) for w in wins]
        M = [gpu.Window(mw, share=W[k], prior=W[k].prior) for k, mw in enumerate(MW)]
        drops = [gpu.margin_old_drops(W[k], MW[k]) for k in range(len(wins))]
        b = gpu.Batch(W, M, drops)
    else:
        b = gpu.Batch(W)
    b.solve(gpu.default_options(iters, fixed, workgroups_per_window=wg))
    if marg:
        b.gauge_fix(); b.marginalize()
    b.synchronize(); b.download_states()
    return W, b, b.summaries()


def _same_bits(sa, sb, Wa, Wb, n):
    for k in range(n):
        assert sa[k].num_iterations == sb[k].num_iterations and sa[k].termination == sb[k].termination, k
        m = min(sa[k].num_iterations, 64)
        for f in ("cost", "cost_candidate", "model_cost_change", "radius", "rho", "step_norm"):
            a = np.array([getattr(sa[k], f)[i] for i in range(m)]); c = np.array([getattr(sb[k], f)[i] for i in range(m)])
            assert np.array_equal(a, c), (k, f, a - c)
        assert sa[k].final_cost == sb[k].final_cost
        for f in ("pose", "sb", "ex", "lam"):
            assert np.array_equal(getattr(Wa[k], f), getattr(Wb[k], f)), (k, f)


def test_cooperative_mode_is_bit_identical_to_one_workgroup_per_window(gpu):
    pre, main, z = golden_windows()
    wins = [main, pre] + [synth.window_at(synth.make_windows(910, 3), k) for k in range(3)]
    Wc, bc, sc = _solve(gpu, wins, 8, True, 0)
    co = bc.cooperative()
    assert co["helpers"] >= 2 and co["groups"] == len(wins) and co["chunks"] >= co["helpers"] and co["last_solve_workgroups"] == 1 + co["helpers"], co
    W1, b1, s1 = _solve(gpu, wins, 8, True, 1)
    assert b1.cooperative() == dict(co, last_solve_workgroups=1)      # the same plan, run by one workgroup per window
    _same_bits(sc, s1, Wc, W1, len(wins))
    # to convergence with the Ceres tolerances (accept / reject and termination decisions included)
    Wc, bc, sc = _solve(gpu, wins, 40, False, 0)
    W1, b1, s1 = _solve(gpu, wins, 40, False, 1)
    _same_bits(sc, s1, Wc, W1, len(wins))
    assert max(s.num_iterations for s in sc) > 9


def test_cooperative_mode_on_a_replay_sized_window_and_more_chunks_than_helpers(gpu):
    """~560 point + 40 line factors (a front-end-sized window): seven helpers; with two helpers forced every helper serves several
    chunks.  Both bit-identical to the one-workgroup run of their own plan; the two plans (different chunkings) agree to rounding."""
    pre, main, z = golden_windows()
    big = synth.window_at(synth.make_windows(77, 1, n_landmarks=140), 0)
    w = dict(main)
    for k in ("proj", "lam"):
        w[k] = big[k]
    res = {}
    try:
        for h in (-1, 2):
            gpu.check(gpu.lib().tcv_set_cooperative(h))
            Wc, bc, sc = _solve(gpu, [w], 8, True, 0, marg=True)
            co = bc.cooperative()
            assert co["helpers"] == (7 if h < 0 else 2) and co["chunks"] >= co["helpers"] and co["last_solve_workgroups"] == 1 + co["helpers"], co
            W1, b1, s1 = _solve(gpu, [w], 8, True, 1, marg=True)
            _same_bits(sc, s1, Wc, W1, 1)
            assert np.array_equal(bc.prior(0).export()["J0"], b1.prior(0).export()["J0"])
            res[h] = (sc[0].final_cost, Wc[0].pose.copy())
    finally:
        gpu.check(gpu.lib().tcv_set_cooperative(-1))
    assert abs(res[-1][0] - res[2][0]) < 1e-9 * res[2][0] and rel(res[-1][1], res[2][1]) < 1e-9


def test_cooperative_mode_off_and_large_batches_keep_one_workgroup_per_window(gpu):
    wins = [synth.window_at(synth.make_windows(920, 2), k) for k in range(2)]
    try:
        gpu.check(gpu.lib().tcv_set_cooperative(0))
        W, b, s = _solve(gpu, wins, 4, True, 0)
        assert b.cooperative()["helpers"] == 0
    finally:
        gpu.check(gpu.lib().tcv_set_cooperative(-1))
    assert gpu.lib().tcv_set_cooperative(9) == gpu.TCV_ERR_INVALID
    many = [synth.window_at(synth.make_windows(930, 130), k) for k in range(130)]      # 130 x (1 + 2) > 256 CUs: no room for helpers
    W, b, s = _solve(gpu, many, 2, True, 0)
    assert b.cooperative()["helpers"] <= 0 or 130 * (1 + b.cooperative()["helpers"]) <= 256


def test_cooperative_launches_in_flight_must_fit_the_chip(gpu):
    """a cooperative kernel spins on its partner workgroups, so all cooperative grids in flight together must be resident: a second
    batch whose groups do not fit next to the first one's runs the SAME plan with one workgroup per window (same bits), and gets its
    groups back once the first batch has been synchronised."""
    n = 24
    wins = [synth.window_at(synth.make_windows(940, n), k) for k in range(n)]
    try:
        gpu.check(gpu.lib().tcv_set_cooperative(7))      # 24 x 8 = 192 of 256 CUs per batch
        Wa = [gpu.Window(w) for w in wins]; Wb = [gpu.Window(w) for w in wins]
        a = gpu.Batch(Wa); b = gpu.Batch(Wb)
        assert a.cooperative()["helpers"] == 7 and b.cooperative()["helpers"] == 7
        o = gpu.default_options(8, True)
        a.solve(o); b.solve(o)                           # a is still in flight (not synchronised) when b is launched
        assert a.cooperative()["last_solve_workgroups"] == 8 and b.cooperative()["last_solve_workgroups"] == 1
        a.synchronize(); b.synchronize(); a.download_states(); b.download_states()
        sa, sb = a.summaries(), b.summaries()
        _same_bits(sa, sb, Wa, Wb, n)
        b.solve(o)
        assert b.cooperative()["last_solve_workgroups"] == 8
        b.synchronize()
    finally:
        gpu.check(gpu.lib().tcv_set_cooperative(-1))


def test_concurrent_one_window_cooperative_launches_spread_over_the_xcds(gpu):
    """the members of a cooperative group share an XCD (workgroup b of a launch is dispatched to XCD b % 8), so the budget a launch is
    admitted against is the XCD's 32 CUs, not the chip's 256: one-window batches with seven helpers each (8 workgroups, a whole CU's LDS
    each) launched on distinct streams without waiting -- the direct batch-API use of many estimators -- are rotated over the XCDs
    (`SolveArgs::coop_rot`: the first group of a launch goes to the XCD with the fewest claimed CUs), 4 per XCD = 32 fit the chip, the
    next ones run the same plan on one workgroup per window.  None waits for partners that cannot be dispatched (status -9 /
    termination FAILURE after the 2 s timeout), all give the bits of the single-workgroup run."""
    hip = C.CDLL("libamdhip64.so")
    n = 36
    win = synth.window_at(synth.make_windows(955, 1), 0)
    o = gpu.default_options(8, True)
    streams = [C.c_void_p() for _ in range(n)]
    try:
        gpu.check(gpu.lib().tcv_set_cooperative(7))
        Wr, br, sr = _solve(gpu, [win], 8, True, 1)
        for st in streams:
            assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0      # hipStreamNonBlocking
        Ws = [[gpu.Window(win)] for _ in range(n)]
        bs = [gpu.Batch(w) for w in Ws]
        assert all(b.cooperative()["helpers"] == 7 for b in bs)
        for b, st in zip(bs, streams):
            b.solve(o, st)                                               # nobody is synchronised before everybody is launched
        wg = [b.cooperative()["last_solve_workgroups"] for b in bs]
        assert wg[:32] == [8] * 32 and wg[32:] == [1] * (n - 32), wg      # 4 launches per XCD are admitted, the rest fall back
        for b, W in zip(bs, Ws):
            b.synchronize(); b.download_states()
            s = b.summaries()
            assert s[0].termination != 5
            _same_bits(s, sr, W, Wr, 1)
        bs[-1].solve(o, streams[-1])                                     # the claims are back after the synchronisation
        assert bs[-1].cooperative()["last_solve_workgroups"] == 8
        bs[-1].synchronize()
    finally:
        gpu.check(gpu.lib().tcv_set_cooperative(-1))
        for st in streams:
            if st:
                hip.hipStreamDestroy(st)
