"""Teacher-forced window-by-window parity for BASELINE configs[3] (round-3 review, item 1b).

A free-running replay compares trajectories: one rounding-level difference, amplified by the window dynamics (the line terms, the
eigenvalue threshold of `marginalization_factor.cpp:284-293`), separates two replays after seconds and says nothing about WHICH window
differed.  Here the ORACLE back end drives the replay (it is the teacher: its results are the ones applied), a spy records every window
it was handed, and afterwards the HIP path gets the same 345 windows of the replay in device batches -- solve, gauge fix
(`estimator.cpp:1537-1581`), marginalisation (MARGIN_OLD `:1911-2046`, MARGIN_SECOND_NEW `:2047-2113`) from IDENTICAL inputs.  Gated per
window: identical iteration count / termination / accept-reject sequence / dogleg cases, states within 1e-6 (north_star), `A'`, `b'`,
`J0'J0`, `J0'r0`, the same MarginalizationInfo layout (m, n, kept blocks, shifted addresses); the NUMBER of thresholded rows of `J0` may
differ only by eigenvalues that are rounding noise (see `compare`); and one window ahead -- window j solved on the HIP-made prior of its
predecessor instead of the oracle's -- positions stay within the north_star's 1 mm.

All five sequences the reference ships ground truth for x {no line factors, given 3D partners, 2D-3D association in the loop
(`estimator.cpp:385-447`, `:671-885`, `:1786-1846`)} -- V2_01_easy included, which the free-running gates with line factors exclude by name
(tests/test_gpu_replay.py: LINE_REPLAY_SEQUENCES)."""
import ctypes as C

import numpy as np
import pytest

import replay
from replay_cache import MODES
from replay_oracle import OracleBackend
from util import fro, rel

pytestmark = pytest.mark.gpu


class Teacher(OracleBackend):
    """the oracle back end of tests/replay_oracle.py, recording per window: the input, the solver summary, the gauge-fixed states and the
    marginalisation's Schur system"""

    def __init__(self):
        self.rec = []

    def optimize(self, win, marg_flag, num_iterations, fixed_iterations):
        import np_oracle as NO
        import orc
        O = orc.Window(win)
        s = O.solve(num_iterations, fixed_iterations)
        st = O.states()
        R0 = NO.q2R(np.asarray(win["pose"])[0, 3:]); P0 = np.asarray(win["pose"])[0, :3]
        Rs, Ps, Vs, po = orc.gauge_fix(R0, P0, st["pose"], st["sb"])
        sb = st["sb"].copy(); sb[:, :3] = Vs
        out = dict(pose=po, sb=sb, ex=st["ex"].copy(), lam=st["lam"].copy(), iterations=s.num_iterations, final_cost=s.final_cost, prior="keep")
        w2 = dict(win, pose=po, speedbias=sb, ex_pose=st["ex"], lam=st["lam"])
        Wn = po.shape[0] - 1
        dbg = None
        if marg_flag == replay.MARGIN_OLD:
            out["prior"], dbg = orc.Window(w2).marginalize_old()
        elif win.get("prior") is not None and ("pose", Wn - 1) in [tuple(b) for b in win["prior"]["blocks"]]:
            prob = NO.Problem(w2)
            out["prior"], dbg = NO.marginalize_second_new(prob, prob.x0())
        n = s.num_iterations
        self.rec.append(dict(win=win, flag=marg_flag, n_it=n, term=s.termination, cost=s.final_cost,
                             ok=[s.step_ok[i] for i in range(min(n, 64))], case=[s.dogleg_case[i] for i in range(min(n, 64))],
                             pose=po, sb=sb, ex=out["ex"], lam=out["lam"], prior=out["prior"] if dbg is not None else None,
                             A=None if dbg is None else np.array(dbg["A_schur"]), b=None if dbg is None else np.array(dbg["b_schur"])))
        return out


def _zero_rows(J0):
    return int((np.abs(np.asarray(J0)).max(axis=1) == 0).sum())


def _producer(rec, j):
    """index of the window whose marginalisation made window j's prior (the replay hands the same dict object on), or None"""
    pj = rec[j]["win"].get("prior")
    if pj is None:
        return None
    for k in range(j - 1, -1, -1):
        if rec[k]["prior"] is pj:
            return k
    return None


def hip_windows(tcv, rec, num_iterations, fixed_iterations, wins=None):
    """every recorded window through the HIP path, batched like replay.HipBackend.optimize_many (windows that marginalise and windows that
    only solve are separate batches).  `wins`: solve these inputs instead of the recorded ones (same structure).  Returns one dict per window."""
    n = len(rec)
    wins = [r["win"] for r in rec] if wins is None else wins
    Ws = [None if w is None else tcv.Window(w) for w in wins]
    Wn = rec[0]["pose"].shape[0] - 1
    out = [None] * n
    for group in (True, False):
        idx = [i for i in range(n) if Ws[i] is not None and (rec[i]["A"] is not None) == group]
        if not idx:
            continue
        if group:
            Ms, drops = [], []
            for i in idx:
                if rec[i]["flag"] == replay.MARGIN_OLD:
                    mw = tcv.margin_old_window(wins[i]); Ms.append(tcv.Window(mw, share=Ws[i], prior=Ws[i].prior)); drops.append(tcv.margin_old_drops(Ws[i], mw))
                else:
                    mw = tcv.margin_second_new_window(wins[i]); Ms.append(tcv.Window(mw, share=Ws[i], prior=Ws[i].prior)); drops.append(tcv.margin_second_new_drops(Ws[i]))
            b = tcv.Batch([Ws[i] for i in idx], Ms, drops)
        else:
            b = tcv.Batch([Ws[i] for i in idx])
        b.solve(tcv.default_options(num_iterations, fixed_iterations)); b.gauge_fix()
        if group:
            b.marginalize()
        b.synchronize(); b.download_states()
        if group:
            b.download_priors()
            status = b.marg_status()
        s = b.summaries()
        for k, i in enumerate(idx):
            o = dict(n_it=s[k].num_iterations, term=s[k].termination, cost=s[k].final_cost,
                     ok=[s[k].step_ok[j] for j in range(min(s[k].num_iterations, 64))], case=[s[k].dogleg_case[j] for j in range(min(s[k].num_iterations, 64))],
                     pose=Ws[i].pose.copy(), sb=Ws[i].sb.copy(), ex=Ws[i].ex.copy(), lam=Ws[i].lam.copy(), status=0, prior=None)
            if group:
                P = b.prior(k)
                d = P.export()
                shift = (lambda nm, j: (nm, j - 1) if nm in ("pose", "sb") else (nm, j)) if rec[i]["flag"] == replay.MARGIN_OLD else \
                        (lambda nm, j: (nm, j - 1) if (nm in ("pose", "sb") and j == Wn) else (nm, j))
                d["blocks"] = tcv.prior_blocks(P, Ws[i], shift)            # getParameterBlocks(addr_shift), marginalization_factor.cpp:301-321
                o.update(dims=P.dims()[:2], schur=P.schur(), prior=d, status=int(status[k]))
            out[i] = o
    return out


def compare(rec, hip):
    """per-window verdicts + the worst figures of the replay"""
    worst = dict(cost=0.0, pose=0.0, sb=0.0, lam=0.0, A=0.0, b=0.0, JtJ=0.0, Jtr=0.0, flip=0.0)
    bad_trace, bad_layout, rows, n_marg = [], [], [], 0
    for k, (r, h) in enumerate(zip(rec, hip)):
        if (h["n_it"], h["term"], h["ok"], h["case"]) != (r["n_it"], r["term"], r["ok"], r["case"]):
            bad_trace.append(k)
            continue
        worst["cost"] = max(worst["cost"], abs(h["cost"] - r["cost"]) / max(abs(r["cost"]), 1e-300))
        worst["pose"] = max(worst["pose"], rel(h["pose"], r["pose"])); worst["sb"] = max(worst["sb"], rel(h["sb"], r["sb"]))
        worst["lam"] = max(worst["lam"], rel(h["lam"], r["lam"]))
        if r["A"] is None:
            continue
        n_marg += 1
        po, ph = r["prior"], h["prior"]
        # MarginalizationInfo layout (marginalization_factor.h:57-70): m, n, keep_block_size / idx, the shifted block addresses, x0
        if (tuple(h["dims"]) != (po["m"], po["n"]) or [tuple(b) for b in ph["blocks"]] != [tuple(b) for b in po["blocks"]]
                or list(ph["sizes"]) != list(po["sizes"]) or list(ph["idx"]) != list(po["idx"])):
            bad_layout.append(k)
            continue
        worst["A"] = max(worst["A"], fro(h["schur"][0], r["A"])); worst["b"] = max(worst["b"], fro(h["schur"][1], r["b"]))
        # the invariants the reference documents (marginalization_factor.cpp:297-298): J0'J0 ~ A', J0'r0 ~ b' -- what the prior IS as a
        # cost function, whichever eigenvectors and signs a decomposition picked
        Jo, Jh = np.asarray(po["J0"]), np.asarray(ph["J0"])
        worst["JtJ"] = max(worst["JtJ"], fro(Jh.T @ Jh, Jo.T @ Jo)); worst["Jtr"] = max(worst["Jtr"], fro(Jh.T @ np.asarray(ph["r0"]), Jo.T @ np.asarray(po["r0"])))
        kh, ko = _zero_rows(Jh), _zero_rows(Jo)
        if kh != ko:
            # eigenvalues of A' are zeroed when <= eps = 1e-8 (marginalization_factor.cpp:284-293); A' is a Schur complement whose terms
            # cancel to ~1e-9 of their size, so its smallest eigenvalues are rounding noise of either sign and two correct FP64
            # evaluations threshold a different number of them.  What must hold: every eigenvalue that one side kept and the other
            # dropped is such noise -- |lambda| tiny against lambda_max
            A = np.asarray(r["A"]); lam = np.linalg.eigvalsh(0.5 * (A + A.T))
            lo, hi = min(kh, ko), max(kh, ko)
            flip = float(np.abs(lam[lo:hi]).max() / lam[-1])
            worst["flip"] = max(worst["flip"], flip)
            rows.append((k, kh, ko, float("%.2g" % flip)))
    return worst, bad_trace, bad_layout, rows, n_marg


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("seq", list(replay.EUROC_SEQUENCES))
def test_teacher_forced_full_length_replay(gpu, seq, mode):
    from replay_cache import teacher_replay
    T = teacher_replay(seq, mode)
    ref, rec = T["ref"], T["rec"]
    assert len(ref["t"]) == len(rec) == 345
    if mode == "given":
        assert min(len(r["win"]["line"]["frame"]) for r in rec[20:]) > 0
    elif mode == "associate":
        assert sum(len(r["win"]["line"]["frame"]) > 0 for r in rec) > 100
    hip = hip_windows(gpu, rec, 8, False)
    worst, bad_trace, bad_layout, rows, n_marg = compare(rec, hip)
    on_net = [k for k, h in enumerate(hip) if h["status"] != 0]
    # one window ahead: window j solved by the HIP path on the HIP-made prior of window k (instead of the oracle's, which the pass above
    # used) against the oracle's solution of window j -- how far ONE solve carries the difference between two correct priors
    wins2, prod = [None] * len(rec), [None] * len(rec)
    for j in range(1, len(rec)):
        k = _producer(rec, j)
        if k is not None and hip[k]["prior"] is not None and k not in bad_layout:
            wins2[j] = dict(rec[j]["win"], prior=hip[k]["prior"]); prod[j] = k
    hip2 = hip_windows(gpu, rec, 8, False, wins2)
    flipped = {k for k, _, _, _ in rows}
    ahead = dict(same=[0.0, 0.0], flipped=[0.0, 0.0])      # [max |dp| metres, max relative cost difference] by whether the producer's row count differed
    ahead_trace = []
    for j, h in enumerate(hip2):
        if h is None:
            continue
        r = rec[j]
        if (h["n_it"], h["term"], h["ok"], h["case"]) != (r["n_it"], r["term"], r["ok"], r["case"]):
            ahead_trace.append(j)
        a = ahead["flipped" if prod[j] in flipped else "same"]
        a[0] = max(a[0], float(np.abs(h["pose"][:, :3] - r["pose"][:, :3]).max())); a[1] = max(a[1], abs(h["cost"] - r["cost"]) / abs(r["cost"]))
    T["rec"] = None                                # (the session cache keeps the trajectory; the window records are large)
    fmt = lambda d: {k: float("%.3g" % v) for k, v in d.items()}
    print("%s / %s: 345 windows (%d marginalised) from identical inputs: traces differ on %s, prior layout on %s, safety net %s; worst %s; "
          "thresholded rows differ on %d windows (window, HIP, oracle, |lambda| / lambda_max of the rows in question): %s; one window ahead on the HIP-made "
          "prior: traces differ on %s, max |dp| [m] / relative cost difference %s"
          % (seq, mode, n_marg, bad_trace, bad_layout, on_net, fmt(worst), len(rows), rows, ahead_trace, {k: [float("%.2g" % x) for x in v] for k, v in ahead.items()}))
    assert not bad_trace and not bad_layout and not on_net, (bad_trace, bad_layout, on_net)
    assert worst["cost"] < 1e-6 and worst["pose"] < 1e-6 and worst["sb"] < 1e-6 and worst["lam"] < 1e-6, worst      # north_star: 1e-6 (measured <= 1e-8)
    assert worst["A"] < 2e-6 and worst["b"] < 2e-6, worst                                                        # measured <= 2e-8 / 1.2e-7
    assert worst["JtJ"] < 2e-6 and worst["Jtr"] < 2e-5, worst                                                     # measured <= 2e-8 / 1.6e-6 (r0 = S^-1/2 V'b: the small retained eigenvalues amplify)
    assert worst["flip"] < 1e-8, rows
    assert len(rows) <= len(rec) // 5, rows
    # north_star: trajectory within 1 mm -- per window, even through a prior whose thresholded rows differ
    assert max(ahead["same"][0], ahead["flipped"][0]) < 1e-3, ahead


# ---- the NATIVE estimator as the teacher (round-4 review, item 5b) ------------------------------------------------------------------
# tests/test_gpu_replay.py compares the native window management (include/tcv_estimator.h) with the Python one free-running: 8 mm over a
# full-length replay, because one rounding-level difference is amplified by the window dynamics.  Here every window the native estimator
# hands to the solver is tapped (tcv_estimator_set_window_tap: parameter arrays, factor lists, pre-integrations, the device-made prior it
# carries from the previous frame) and re-solved by the CPU oracle from those identical inputs: per window, whatever the trajectory did.
_dp, _ip = C.POINTER(C.c_double), C.POINTER(C.c_int)


class _Snapshot(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("n_frames", "n_landmarks", "n_imu", "n_proj", "n_line", "marg_flag", "estimate_extrinsic", "line_exact_jacobian")] + \
               [(k, _dp) for k in ("pose_in", "speedbias_in", "ex_pose_in", "feature_in", "pose_out", "speedbias_out", "ex_pose_out", "feature_out")] + \
               [("imu", C.c_void_p), ("imu_frame_i", _ip), ("imu_frame_j", _ip), ("proj_frame_i", _ip), ("proj_frame_j", _ip), ("proj_feature", _ip), ("proj_pts", _dp),
                ("line_frame", _ip), ("line_data", _dp), ("line_K", C.c_double * 9), ("line_Ric", C.c_double * 9), ("line_Tic", C.c_double * 3), ("gravity", C.c_double * 3),
                ("proj_sqrt_info", C.c_double), ("prior_m", C.c_int), ("prior_n", C.c_int), ("prior_nblk", C.c_int),
                ("prior_block_kind", _ip), ("prior_block_index", _ip), ("prior_block_size", _ip), ("prior_block_idx", _ip),
                ("prior_x0", _dp), ("prior_J0", _dp), ("prior_r0", _dp), ("iterations", C.c_int), ("applied", C.c_int), ("final_cost", C.c_double)]


def _snapshot_window(tcv, S):
    """tcv_window_snapshot -> the window dict the oracle (and tcv.Window) take, and the native results"""
    arr = lambda p, n, shape=None: np.ctypeslib.as_array(p, shape=(n,)).copy().reshape(shape or (n,)) if n else np.zeros(shape or (0,))
    iarr = lambda p, n: np.ctypeslib.as_array(p, shape=(n,)).astype(np.int64).copy() if n else np.zeros(0, np.int64)
    F, L, NI, NP, NL = S.n_frames, S.n_landmarks, S.n_imu, S.n_proj, S.n_line
    pre = (tcv.ImuPreintegration * max(1, NI)).from_address(S.imu) if NI else []
    imu = {"delta_p": np.array([list(pre[k].delta_p) for k in range(NI)]).reshape(NI, 3), "delta_q": np.array([list(pre[k].delta_q) for k in range(NI)]).reshape(NI, 4),
           "delta_v": np.array([list(pre[k].delta_v) for k in range(NI)]).reshape(NI, 3), "lin_ba": np.array([list(pre[k].linearized_ba) for k in range(NI)]).reshape(NI, 3),
           "lin_bg": np.array([list(pre[k].linearized_bg) for k in range(NI)]).reshape(NI, 3), "sum_dt": np.array([pre[k].sum_dt for k in range(NI)]),
           "jacobian": np.array([list(pre[k].jacobian) for k in range(NI)]).reshape(NI, 15, 15), "covariance": np.array([list(pre[k].covariance) for k in range(NI)]).reshape(NI, 15, 15),
           "frame_i": iarr(S.imu_frame_i, NI), "frame_j": iarr(S.imu_frame_j, NI)}
    pts = arr(S.proj_pts, 6 * NP, (NP, 6))
    proj = dict(frame_i=iarr(S.proj_frame_i, NP), frame_j=iarr(S.proj_frame_j, NP), landmark=iarr(S.proj_feature, NP), pts_i=pts[:, :3].copy(), pts_j=pts[:, 3:].copy(),
                sqrt_info=S.proj_sqrt_info, loss_a=1.0)
    ld = arr(S.line_data, 9 * NL, (NL, 9))
    line = dict(frame=iarr(S.line_frame, NL), pts_start=ld[:, :3].copy(), pts_end=ld[:, 3:6].copy(), abc=ld[:, 6:].copy(), K=np.array(list(S.line_K)).reshape(3, 3),
                Ric=np.array(list(S.line_Ric)).reshape(3, 3), Tic=np.array(list(S.line_Tic)), loss_a=1.0, exact_jacobian=bool(S.line_exact_jacobian))
    prior = None
    if S.prior_n > 0:
        nb, n = S.prior_nblk, S.prior_n
        sizes = [int(v) for v in iarr(S.prior_block_size, nb)]
        x0 = arr(S.prior_x0, sum(sizes))
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
        kinds = {0: "pose", 1: "sb", 2: "ex"}
        prior = dict(m=S.prior_m, n=n, sizes=sizes, idx=[int(v) - S.prior_m for v in iarr(S.prior_block_idx, nb)], x0=[x0[offs[k]:offs[k + 1]].copy() for k in range(nb)],
                     J0=arr(S.prior_J0, n * n, (n, n)).T.copy(), r0=arr(S.prior_r0, n),          # (column-major on the wire)
                     blocks=[(kinds[int(k)], int(i)) for k, i in zip(iarr(S.prior_block_kind, nb), iarr(S.prior_block_index, nb))])
    win = dict(pose=arr(S.pose_in, 7 * F, (F, 7)), speedbias=arr(S.speedbias_in, 9 * F, (F, 9)), ex_pose=arr(S.ex_pose_in, 7), lam=arr(S.feature_in, L),
               imu=imu, proj=proj, line=line, G=np.array(list(S.gravity)), prior=prior)
    res = dict(pose=arr(S.pose_out, 7 * F, (F, 7)), sb=arr(S.speedbias_out, 9 * F, (F, 9)), ex=arr(S.ex_pose_out, 7), lam=arr(S.feature_out, L),
               iterations=S.iterations, cost=S.final_cost, applied=S.applied, flag=S.marg_flag)
    return win, res


@pytest.mark.parametrize("seq,mode,extrinsic", [("V1_02_medium", "associate", True), ("V2_01_easy", "given", True), ("V2_03_difficult", "none", True),
                                                 ("V1_03_difficult", "given", False)])      # False: ESTIMATE_EXTRINSIC = 0 (config/*/…_config.yaml of several rigs): para_Ex_Pose constant in the solve, kept in the prior
def test_native_estimator_windows_resolved_by_the_oracle(gpu, seq, mode, extrinsic):
    """120 frames of a native lock-step replay; every window it optimises -- from its own states, its own device-resident pre-integrations and the
    device-made prior of its previous frame -- is solved again by the C oracle: iterations, final cost and the gauge-fixed states within the
    north_star's 1e-6, window by window"""
    import np_oracle as NO
    import orc
    tcv = gpu
    st = replay.simulate_stream_euroc(seq, 120, start_s=0.5, max_features=60, **MODES[mode])
    ls = replay.NativeLockstep([st], num_iterations=8, estimate_extrinsic=extrinsic)
    L = tcv.lib()
    L.tcv_estimator_set_window_tap.argtypes = [C.c_void_p, C.c_int]
    L.tcv_estimator_get_window_snapshot.argtypes = [C.c_void_p, C.POINTER(_Snapshot)]
    tcv.check(L.tcv_estimator_set_window_tap(ls.ests[0], 1))
    worst = dict(cost=0.0, pose=0.0, sb=0.0, lam=0.0, ex=0.0)
    n_win, n_prior, bad_it = 0, 0, []
    try:
        for k in range(ls.n_frames):
            if not ls.step(k):
                continue
            S = _Snapshot()
            tcv.check(L.tcv_estimator_get_window_snapshot(ls.ests[0], C.byref(S)))
            win, res = _snapshot_window(tcv, S)
            assert res["applied"] == 1 and bool(S.estimate_extrinsic) == extrinsic
            O = orc.Window(win, ex_constant=not extrinsic)
            so = O.solve(8, False)
            sto = O.states()
            R0 = NO.q2R(win["pose"][0, 3:]); P0 = win["pose"][0, :3]
            Rs, Ps, Vs, po = orc.gauge_fix(R0, P0, sto["pose"], sto["sb"])
            sbo = sto["sb"].copy(); sbo[:, :3] = Vs
            n_win += 1; n_prior += win["prior"] is not None
            if so.num_iterations != res["iterations"]:
                bad_it.append((k, res["iterations"], so.num_iterations))
                continue
            worst["cost"] = max(worst["cost"], abs(res["cost"] - so.final_cost) / abs(so.final_cost))
            worst["pose"] = max(worst["pose"], rel(res["pose"], po)); worst["sb"] = max(worst["sb"], rel(res["sb"], sbo))
            worst["lam"] = max(worst["lam"], rel(res["lam"], sto["lam"])); worst["ex"] = max(worst["ex"], rel(res["ex"], sto["ex"]))
    finally:
        ls.close()
    print("%s / %s: %d native windows (%d on a device-made prior) re-solved by the oracle: iteration counts differ on %s; worst %s"
          % (seq, mode, n_win, n_prior, bad_it, {k: float("%.3g" % v) for k, v in worst.items()}))
    assert n_win >= 100 and n_prior >= 95
    assert not bad_it, bad_it
    assert all(v < 1e-6 for v in worst.values()), worst


def test_teacher_forced_replay_through_a_standstill(gpu):
    """A platform at rest for 11.5 s (tests/replay_cache.py STANDSTILL): 115 MARGIN_SECOND_NEW frames in a row, then windows whose newest IMU
    factor is left out (sum_dt > 10, estimator.cpp:1726), then ten keyframes with the gap in the MIDDLE of the window -- two IMU chains next to a
    prior, the case that caught the chain elimination in round 5 -- and a MARGIN_OLD without its IMU factor (:1933).  Every window the oracle
    replay solved goes through the HIP path from identical inputs: identical traces and prior layouts, states within 1e-6."""
    from replay_cache import STANDSTILL, teacher_replay
    T = teacher_replay(STANDSTILL, "given")
    rec = T["rec"]
    n_imu = [len(r["win"]["imu"]["sum_dt"]) for r in rec]
    flags = [r["flag"] for r in rec]
    gap_inside = [k for k, r in enumerate(rec) if len(r["win"]["imu"]["sum_dt"]) == 9 and list(np.asarray(r["win"]["imu"]["frame_i"])) != list(range(9))]
    print("standstill: %d windows, %d without one IMU factor (%d of them with the gap inside the window), %d MARGIN_SECOND_NEW"
          % (len(rec), sum(v == 9 for v in n_imu), len(gap_inside), flags.count(replay.MARGIN_SECOND_NEW)))
    assert len(rec) >= 170 and sum(v == 9 for v in n_imu) >= 15 and len(gap_inside) >= 8 and flags.count(replay.MARGIN_SECOND_NEW) >= 100
    hip = hip_windows(gpu, rec, 8, False)
    worst, bad_trace, bad_layout, rows, n_marg = compare(rec, hip)
    print("standstill: traces differ on %s, layouts on %s; worst %s; thresholded rows differ on %d windows" % (bad_trace, bad_layout, {k: float("%.3g" % v) for k, v in worst.items()}, len(rows)))
    T["rec"] = None
    assert not bad_trace and not bad_layout, (bad_trace, bad_layout)
    assert worst["cost"] < 1e-6 and worst["pose"] < 1e-6 and worst["sb"] < 1e-6 and worst["lam"] < 1e-6, worst
    assert worst["A"] < 2e-6 and worst["b"] < 2e-6 and worst["JtJ"] < 2e-6, worst
    assert worst["flip"] < 1e-8, rows


def test_native_estimator_through_a_standstill(gpu):
    """the same stream through the native window management (include/tcv_estimator.h): every window it builds while the platform rests and
    after it moves on -- IMU buffers merged frame after frame, the long pre-integration left out, the gap travelling through the window -- is
    re-solved by the oracle from the tapped inputs"""
    import np_oracle as NO
    import orc
    from replay_cache import STANDSTILL, stream_of
    tcv = gpu
    st = stream_of(STANDSTILL, "given")
    ls = replay.NativeLockstep([st], num_iterations=8)
    L = tcv.lib()
    L.tcv_estimator_set_window_tap.argtypes = [C.c_void_p, C.c_int]
    L.tcv_estimator_get_window_snapshot.argtypes = [C.c_void_p, C.POINTER(_Snapshot)]
    tcv.check(L.tcv_estimator_set_window_tap(ls.ests[0], 1))
    worst = dict(cost=0.0, pose=0.0, sb=0.0, lam=0.0, ex=0.0)
    n_win, n_gap, bad_it = 0, 0, []
    try:
        for k in range(ls.n_frames):
            if not ls.step(k):
                continue
            S = _Snapshot()
            tcv.check(L.tcv_estimator_get_window_snapshot(ls.ests[0], C.byref(S)))
            win, res = _snapshot_window(tcv, S)
            assert res["applied"] == 1
            O = orc.Window(win); so = O.solve(8, False); sto = O.states()
            Rs, Ps, Vs, po = orc.gauge_fix(NO.q2R(win["pose"][0, 3:]), win["pose"][0, :3], sto["pose"], sto["sb"])
            sbo = sto["sb"].copy(); sbo[:, :3] = Vs
            n_win += 1; n_gap += int(len(win["imu"]["sum_dt"]) < replay.WINDOW_SIZE)      # (the window holds the factors that passed the 10 s rule)
            if so.num_iterations != res["iterations"]:
                bad_it.append((k, res["iterations"], so.num_iterations))
                continue
            worst["cost"] = max(worst["cost"], abs(res["cost"] - so.final_cost) / abs(so.final_cost))
            worst["pose"] = max(worst["pose"], rel(res["pose"], po)); worst["sb"] = max(worst["sb"], rel(res["sb"], sbo))
            worst["lam"] = max(worst["lam"], rel(res["lam"], sto["lam"])); worst["ex"] = max(worst["ex"], rel(res["ex"], sto["ex"]))
    finally:
        ls.close()
    print("standstill, native: %d windows (%d without the IMU factor of an interval over 10 s), iteration counts differ on %s; worst %s" % (n_win, n_gap, bad_it, {k: float("%.3g" % v) for k, v in worst.items()}))
    assert n_win >= 170 and n_gap >= 15 and not bad_it, (n_win, n_gap, bad_it)
    assert all(v < 1e-6 for v in worst.values()), worst
