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
