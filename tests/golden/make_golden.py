"""Generates the committed golden vectors under tests/golden/ (run in the authoring container):

    python tests/golden/make_golden.py

The reference (ZHENGXi-git/TC-VIML) ships no tests or known-answer vectors for this path and cannot be
built here (Eigen/Ceres/ROS absent), so these vectors come from the NumPy restatement oracle/np_oracle.py
(parity unpinned).  They pin: the C oracle, the HIP kernels, and any future refactor of
either against the numbers this script produced.  Inputs are stored next to the expected outputs.
"""
import os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd"))
import np_oracle as npo
import synth


def factor_cases(seed=20251001, n=32):
    rng = np.random.default_rng(seed)
    batch = synth.make_windows(1000, 4)
    out = {}
    # ---- P1 projection factor: n cases drawn from synthetic windows + perturbed states
    pi, pj, ex, lam, pti, ptj, r, J0, J1, J2, J3 = ([] for _ in range(11))
    for k in range(n):
        w = synth.window_at(batch, k % 4); pr = w["proj"]
        f = int(rng.integers(0, len(pr["frame_i"])))
        a = w["pose"][pr["frame_i"][f]].copy(); b = w["pose"][pr["frame_j"][f]].copy(); e = w["ex_pose"].copy()
        a[:3] += 0.05 * rng.normal(size=3); b[:3] += 0.05 * rng.normal(size=3)
        if k % 5 == 0:      # slightly non-unit quaternions exercise the un-normalised Eigen semantics
            a[3:] *= 1.0 + 1e-3 * rng.normal(); b[3:] *= 1.0 + 1e-3 * rng.normal()
        l = float(w["lam"][pr["landmark"][f]]) * (1 + 0.2 * rng.normal())
        rr, JJ = npo.proj_evaluate(a, b, e, l, pr["pts_i"][f], pr["pts_j"][f], pr["sqrt_info"])
        pi.append(a); pj.append(b); ex.append(e); lam.append(l); pti.append(pr["pts_i"][f]); ptj.append(pr["pts_j"][f])
        r.append(rr); J0.append(JJ[0]); J1.append(JJ[1]); J2.append(JJ[2]); J3.append(JJ[3])
    out.update(p1_pose_i=np.array(pi), p1_pose_j=np.array(pj), p1_ex=np.array(ex), p1_lam=np.array(lam), p1_pts_i=np.array(pti),
               p1_pts_j=np.array(ptj), p1_sqrt_info=np.array(synth.PROJ_SQRT_INFO), p1_r=np.array(r), p1_J0=np.array(J0),
               p1_J1=np.array(J1), p1_J2=np.array(J2), p1_J3=np.array(J3))
    # ---- L1 line factor
    po, ps, pe, abc, r, J = ([] for _ in range(6))
    for k in range(n):
        w = synth.window_at(batch, k % 4); ln = w["line"]
        f = int(rng.integers(0, len(ln["frame"])))
        a = w["pose"][ln["frame"][f]].copy(); a[:3] += 0.03 * rng.normal(size=3)
        if k % 4 == 0: a[3:] *= 1.0 + 1e-2 * rng.normal()      # line factor normalises the quaternion itself (:33)
        rr, JJ = npo.line_evaluate(a, ln["pts_start"][f], ln["pts_end"][f], ln["abc"][f], ln["K"], ln["Ric"], ln["Tic"])
        po.append(a); ps.append(ln["pts_start"][f]); pe.append(ln["pts_end"][f]); abc.append(ln["abc"][f]); r.append(rr); J.append(JJ[0])
    w = synth.window_at(batch, 0)
    out.update(l1_pose=np.array(po), l1_start=np.array(ps), l1_end=np.array(pe), l1_abc=np.array(abc), l1_K=w["line"]["K"],
               l1_Ric=w["line"]["Ric"], l1_Tic=w["line"]["Tic"], l1_r=np.array(r), l1_J=np.array(J))
    # ---- I1 IMU factor (sqrt_info pinned with the inputs: SURVEY.md Appendix A)
    keys = ["delta_p", "delta_q", "delta_v", "lin_ba", "lin_bg", "sum_dt", "jacobian", "covariance"]
    acc = {k: [] for k in keys}
    pi, si, pj, sj, sq, r, Js = [], [], [], [], [], [], [[], [], [], []]
    for k in range(n):
        w = synth.window_at(batch, k % 4); im = w["imu"]
        f = int(rng.integers(0, len(im["frame_i"])))
        i, j = int(im["frame_i"][f]), int(im["frame_j"][f])
        a = w["pose"][i].copy(); b = w["pose"][j].copy(); sa = w["speedbias"][i].copy(); sb = w["speedbias"][j].copy()
        sa[3:] += 0.01 * rng.normal(size=6); sb[3:] += 0.01 * rng.normal(size=6)       # non-zero bias corrections
        pre = {kk: im[kk][f] for kk in keys}
        pre["sum_dt"] = float(pre["sum_dt"])
        S = npo.imu_sqrt_info(pre["covariance"])
        rr, JJ = npo.imu_evaluate(a, sa, b, sb, pre, w["G"], sqrt_info=S)
        for kk in keys: acc[kk].append(im[kk][f])
        pi.append(a); si.append(sa); pj.append(b); sj.append(sb); sq.append(S); r.append(rr)
        for q in range(4): Js[q].append(JJ[q])
    out.update({f"i1_{k}": np.array(v) for k, v in acc.items()})
    out.update(i1_pose_i=np.array(pi), i1_sb_i=np.array(si), i1_pose_j=np.array(pj), i1_sb_j=np.array(sj), i1_sqrt_info=np.array(sq),
               i1_G=w["G"], i1_r=np.array(r), i1_J0=np.array(Js[0]), i1_J1=np.array(Js[1]), i1_J2=np.array(Js[2]), i1_J3=np.array(Js[3]))
    # ---- S2 Plus and C1 corrector
    x = np.array([w["pose"][k % 11] for k in range(n)]); d = 0.05 * rng.normal(size=(n, 6)); d[0] = 0.0
    out.update(s2_x=x, s2_delta=d, s2_out=np.array([npo.pose_plus(x[k], d[k]) for k in range(n)]))
    rs = rng.normal(size=(n, 2)) * np.logspace(-3, 3, n)[:, None]; rs[0] = 0.0
    Jc = rng.normal(size=(n, 2, 7))
    cr, cJ, cc = [], [], []
    for k in range(n):
        a, b, c = npo.loss_correct(rs[k].copy(), [Jc[k].copy()], 1.0)
        cr.append(a); cJ.append(b[0]); cc.append(c)
    out.update(c1_r=rs, c1_J=Jc, c1_r_out=np.array(cr), c1_J_out=np.array(cJ), c1_cost=np.array(cc))
    return out


def flatten_window(w, prefix):
    d = {}
    for k in ("pose", "speedbias", "ex_pose", "lam", "G"): d[prefix + k] = np.asarray(w[k])
    for grp in ("imu", "proj", "line"):
        for k, v in w[grp].items():
            if k in ("acc", "gyr"): continue
            d[f"{prefix}{grp}_{k}"] = np.asarray(v if v is not None else 0.0)
    return d


def unflatten_window(z, prefix):
    w = {k: z[prefix + k] for k in ("pose", "speedbias", "ex_pose", "lam", "G")}
    for grp in ("imu", "proj", "line"):
        p = f"{prefix}{grp}_"
        w[grp] = {k[len(p):]: z[k] for k in z.files if k.startswith(p)}
    for k in ("sqrt_info", "loss_a"): w["proj"][k] = float(w["proj"][k])
    w["line"]["loss_a"] = float(w["line"]["loss_a"])
    w["prior"] = None
    return w


def window_case(wid=77):
    """pre-window -> solve(8) -> MARGIN_OLD prior -> main window with prior -> solve(8): everything by the NumPy oracle."""
    out = {}
    pre = synth.window_at(synth.make_windows(wid, 1, frame_shift=-1), 0)
    main = synth.window_at(synth.make_windows(wid, 1), 0)
    out.update(flatten_window(pre, "pre_")); out.update(flatten_window(main, "main_"))
    prob = npo.Problem(pre)
    sq = [npo.imu_sqrt_info(pre["imu"]["covariance"][k]) for k in range(10)]
    x, s = npo.solve(prob, 8, True, imu_sqrt=sq)
    out["pre_cost"] = np.array([it["cost"] for it in s["iterations"]])
    out["pre_final_pose"] = x["pose"]; out["pre_final_sb"] = x["sb"]; out["pre_final_ex"] = x["ex"]; out["pre_final_lam"] = x["lam"]
    out["pre_first_delta"] = s["iterations"][1]["delta"]
    out["pre_model_cost_change"] = np.array([it.get("model_cost_change", 0.0) for it in s["iterations"]])
    out["pre_case"] = np.array([it.get("case", 0) for it in s["iterations"]])
    prior, dbg = npo.marginalize_old(prob, x, imu_sqrt=sq)
    out.update(marg_m=np.array(prior["m"]), marg_n=np.array(prior["n"]), marg_A_schur=dbg["A_schur"], marg_b_schur=dbg["b_schur"],
               marg_J0=prior["J0"], marg_r0=prior["r0"], marg_sizes=np.array(prior["sizes"]), marg_idx=np.array(prior["idx"]),
               marg_x0=np.concatenate([np.atleast_1d(v) for v in prior["x0"]]),
               marg_block_kind=np.array([{"pose": 0, "sb": 1, "ex": 2}[b[0]] for b in prior["blocks"]]),
               marg_block_index=np.array([b[1] for b in prior["blocks"]]))
    main["prior"] = prior
    prob2 = npo.Problem(main)
    sq2 = [npo.imu_sqrt_info(main["imu"]["covariance"][k]) for k in range(10)]
    x2, s2 = npo.solve(prob2, 8, True, imu_sqrt=sq2)
    out["main_cost"] = np.array([it["cost"] for it in s2["iterations"]])
    out["main_final_pose"] = x2["pose"]; out["main_final_sb"] = x2["sb"]; out["main_final_ex"] = x2["ex"]; out["main_final_lam"] = x2["lam"]
    out["main_first_delta"] = s2["iterations"][1]["delta"]
    out["main_model_cost_change"] = np.array([it.get("model_cost_change", 0.0) for it in s2["iterations"]])
    out["main_case"] = np.array([it.get("case", 0) for it in s2["iterations"]])
    # prior residual / Jacobian-product known answers at the initial main state (M0)
    blocks_x = [npo.Problem.get(prob2.x0(), nm, i) for (nm, i) in prior["blocks"]]
    r, _ = npo.prior_evaluate(prior, blocks_x, want_jac=False)
    out["m0_r"] = r
    # convergence-mode run (Ceres tolerances on)
    x3, s3 = npo.solve(prob2, 50, False, imu_sqrt=sq2)
    out["conv_num_iterations"] = np.array(len(s3["iterations"])); out["conv_final_cost"] = np.array(s3["final_cost"])
    out["conv_termination"] = np.array(["NO_CONVERGENCE", "CONVERGENCE_GRADIENT", "CONVERGENCE_PARAMETER", "CONVERGENCE_FUNCTION",
                                        "CONVERGENCE_RADIUS", "FAILURE"].index(s3["termination"]))
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "factors.npz"), **factor_cases())
    np.savez_compressed(os.path.join(HERE, "window.npz"), **window_case())
    for f in ("factors.npz", "window.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
