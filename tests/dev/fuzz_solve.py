#!/usr/bin/env python3
"""Developer checker (GPU): randomised window STRUCTURES through the fused solve against the C oracle (oracle/ is the checker here, as in
tests/).  Every case is a seeded mutation of a golden or synthetic window -- ragged tracks, landmarks seen once, few or no points / lines,
short windows, IMU factors left out (sum_dt > 10 s), constant extrinsics, with and without the prior -- solved four ways on the device:

    lone      a batch of one (cooperative kernels when the window is large enough for helpers, 160 KiB of LDS)
    single    a batch of one with TCV_COOP_H=0 (one workgroup, 160 KiB)
    packed    257 copies (a batch larger than the chip: two workgroups per CU, 80 KiB each)
    dense     tcv_set_solver_variant(1): the dense layout

and compared with the oracle: first step (no trust-region decision behind it) to 1e-6, then trace, final cost and states.  A window whose
trust-region decisions are borderline may legitimately take another branch; such cases are listed as `trace` and their first step still
has to agree.  Mutations produce ill-posed windows too (a pose seen by one feature and no IMU factor, landmarks seen once at no parallax, two
frames without a prior): there the answer is set by rounding in directions only the trust region's mu D^2 holds.  The checker measures
that on the ORACLE ALONE -- the same window with its states moved by 1e-13 relative, three draws -- and allows the device 30x the oracle's
own movement where that exceeds 1e-6 (`ok~`: listed with the measured sensitivity).

    python tests/dev/fuzz_solve.py [cases] [first seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc      # noqa: E402
import synth    # noqa: E402
import tcv      # noqa: E402
from util import fro, golden_windows, rel, sub_window      # noqa: E402

TOL = 1e-6


def take(d, keep, n):
    return {k: (np.asarray(v)[keep] if isinstance(v, np.ndarray) and np.asarray(v).shape[:1] == (n,) else v) for k, v in d.items()}


def thin_points(rng, w, mode):
    pr = {k: np.asarray(v) if isinstance(v, (list, np.ndarray)) else v for k, v in w["proj"].items()}
    n = len(pr["landmark"])
    if n == 0:
        return w
    keep = np.ones(n, bool)
    lm = pr["landmark"]
    for l in np.unique(lm):
        idx = np.nonzero(lm == l)[0]
        if mode == "prefix":
            keep[idx[int(rng.integers(1, len(idx) + 1)):]] = False
        elif mode == "once":
            keep[idx[1:]] = False
        elif mode == "random":
            k = rng.random(len(idx)) < 0.5
            if not k.any():
                k[int(rng.integers(len(idx)))] = True
            keep[idx] = k
        elif mode == "few":      # most landmarks vanish altogether
            if rng.random() < 0.8:
                keep[idx] = False
    out = dict(w)
    out["proj"] = take(pr, keep, n)
    used = sorted(set(int(l) for l in out["proj"]["landmark"]))
    remap = {l: i for i, l in enumerate(used)}
    out["proj"]["landmark"] = np.array([remap[int(l)] for l in out["proj"]["landmark"]], int)
    out["lam"] = np.asarray(w["lam"])[used] if used else np.zeros(0)
    return out


def thin_lines(rng, w, frac):
    ln = w["line"]
    n = len(ln["frame"])
    keep = rng.random(n) < frac
    out = dict(w)
    out["line"] = take(ln, keep, n)
    return out


def make_case(rng, seed):
    pre, main, z = golden_windows()
    # (seeds from 100000 on also draw `chained`: a synthetic window on the prior the ORACLE makes from the window one frame earlier -- priors of
    # many layouts next to every mutation; the list below 100000 is unchanged so that the recorded seed ranges keep their cases)
    kind = rng.choice(["main", "pre", "synth", "synth", "short"] + (["chained", "chained", "chained"] if seed >= 100000 else []))
    notes = [str(kind)]
    if kind == "main":
        w = dict(main)
    elif kind == "pre":
        w = dict(pre)
    elif kind == "chained":
        nl = int(rng.choice([3, 17, 50, 120, 200])); nn = int(rng.choice([1, 7, 40, 100]))
        before = dict(synth.window_at(synth.make_windows(40000 + seed, 1, n_landmarks=nl, n_lines=nn, frame_shift=-1), 0), prior=None)
        Op = orc.Window(before); Op.solve(8, True); st = Op.states()
        before = dict(before, pose=st["pose"], speedbias=st["sb"], ex_pose=st["ex"], lam=st["lam"])
        po, _ = orc.Window(before).marginalize_old()
        w = dict(synth.window_at(synth.make_windows(40000 + seed, 1, n_landmarks=nl, n_lines=nn), 0), prior=po)
        notes.append(f"L{nl} n{nn} prior n={po['n']}")
    else:
        nl = int(rng.choice([1, 3, 17, 50, 120, 200, 290]))
        nn = int(rng.choice([0, 1, 7, 40, 100]))
        w = dict(synth.window_at(synth.make_windows(40000 + seed, 1, n_landmarks=nl, n_lines=max(nn, 1)), 0), prior=None)
        if nn == 0:
            w = thin_lines(rng, w, 0.0)
        notes.append(f"L{nl} n{nn}")
        if kind == "short":
            f = int(rng.integers(2, 11))
            w = sub_window(w, f)
            notes.append(f"frames {f}")
    mode = str(rng.choice(["full", "prefix", "once", "random", "few"]))
    if mode != "full":
        w = thin_points(rng, w, mode)
    notes.append(mode)
    if rng.random() < 0.4:
        fr = float(rng.choice([0.0, 0.1, 0.5]))
        w = thin_lines(rng, w, fr)
        notes.append(f"lines x{fr}")
    gone = []
    nimu = len(w["imu"]["sum_dt"])
    if nimu > 1 and rng.random() < 0.35:
        gone = sorted(set(int(g) for g in rng.integers(0, nimu, size=int(rng.integers(1, 3)))))
        notes.append(f"imu gone {gone}")
    exc = bool(rng.random() < 0.25)
    if exc:
        notes.append("ex const")
    w_hip, w_orc = w, w
    if gone:
        im = dict(w["imu"])
        sd = np.array(im["sum_dt"], dtype=float).copy(); sd[gone] = 11.0; im["sum_dt"] = sd
        w_hip = dict(w, imu=im)
        keep = ~np.isin(np.arange(nimu), gone)
        w_orc = dict(w, imu=take(w["imu"], keep, nimu))
    return w_hip, w_orc, exc, " ".join(notes)


def oracle_sensitivity(w_orc, exc, so0, st0):
    """how far the oracle's own first step / final cost / states move when the window's states move by 1e-13 relative"""
    f0 = np.array(so0.first_delta[:so0.n_local])
    sens = dict(first=0.0, cost=0.0, pose=0.0, sb=0.0, ex=0.0, lam=0.0)
    for rep in range(3):
        r2 = np.random.Generator(np.random.PCG64(977 + rep))
        w2 = dict(w_orc)
        for key in ("lam", "pose", "speedbias"):
            a = np.asarray(w_orc[key], dtype=float)
            w2[key] = a * (1 + 1e-13 * r2.standard_normal(a.shape))
        O = orc.Window(w2, ex_constant=exc); so = O.solve(8, True)
        f1 = np.array(so.first_delta[:so.n_local])
        if len(f0):
            sens["first"] = max(sens["first"], fro(f1, f0))
        sens["cost"] = max(sens["cost"], abs(so.final_cost - so0.final_cost) / max(so0.final_cost, 1e-12))
        st = O.states()
        for key in ("pose", "sb", "ex", "lam"):
            sens[key] = max(sens[key], rel(st[key], st0[key]))
    return sens


def compare(W, b, s, k, O, so, sens):
    """('ok' | 'ok~' | 'trace' | 'FIRST STEP' | 'STATE', detail)"""
    fo = np.array(so.first_delta[:so.n_local]); fg = b.first_step(k)
    if len(fg) != len(fo):
        return "FIRST STEP", f"length {len(fg)} vs {len(fo)}"
    soft = []
    d1 = fro(fg, fo) if len(fo) else 0.0
    if not d1 < TOL:
        if not d1 < 30 * sens["first"]:
            return "FIRST STEP", f"first step differs {d1:.2e} (oracle moves {sens['first']:.1e})"
        soft.append(f"first step {d1:.1e} / oracle {sens['first']:.1e}")
    n = so.num_iterations
    same = s.num_iterations == n and s.termination == so.termination and \
        [s.dogleg_case[i] for i in range(1, n)] == [so.dogleg_case[i] for i in range(1, n)] and [s.step_ok[i] for i in range(1, n)] == [so.step_ok[i] for i in range(1, n)]
    if not same:
        return "trace", f"iterations {s.num_iterations} vs {n}"
    dc = abs(s.final_cost - so.final_cost) / max(so.final_cost, 1e-12)      # (a zero-residual window: absolute below 1e-12)
    if not dc <= TOL:
        if not dc < 30 * sens["cost"]:
            return "STATE", f"final cost {s.final_cost:.9e} vs {so.final_cost:.9e} (oracle moves {sens['cost']:.1e})"
        soft.append(f"cost {dc:.1e} / oracle {sens['cost']:.1e}")
    st, sg = O.states(), W.states()
    for key in ("pose", "sb", "ex", "lam"):
        r = rel(sg[key], st[key])
        if not r < TOL:
            if not r < 30 * sens[key]:
                return "STATE", f"{key} differs {r:.2e} (oracle moves {sens[key]:.1e})"
            soft.append(f"{key} {r:.1e} / oracle {sens[key]:.1e}")
    return ("ok~", "; ".join(soft)) if soft else ("ok", "")


def gpu_run(w, exc, copies, coop_off=False, dense=False, mfma=True):
    L = tcv.lib()
    old = os.environ.get("TCV_COOP_H")
    if coop_off:
        os.environ["TCV_COOP_H"] = "0"
    if dense:
        L.tcv_set_solver_variant(1)
    try:
        Ws = [tcv.Window(w, estimate_extrinsic=not exc) for _ in range(copies)]
        b = tcv.Batch(Ws)
        b.solve(tcv.default_options(8, True, mfma, 256, True))
        b.synchronize(); b.download_states()
        return Ws, b, b.summaries()
    finally:
        if dense:
            L.tcv_set_solver_variant(0)
        if coop_off:
            if old is None:
                os.environ.pop("TCV_COOP_H", None)
            else:
                os.environ["TCV_COOP_H"] = old


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    tally = {}
    bad = []
    kept = []
    for c in range(cases):
        seed = seed0 + c
        rng = np.random.Generator(np.random.PCG64(seed))
        w_hip, w_orc, exc, note = make_case(rng, seed)
        try:
            O = orc.Window(w_orc, ex_constant=exc)
            so = O.solve(8, True)
            sens = oracle_sensitivity(w_orc, exc, so, O.states())
        except Exception as e:      # noqa: BLE001
            print(f"case {seed} [{note}]: oracle refused: {e}")
            continue
        kept.append((seed, note, w_hip, exc, O, so, sens))
        row = []
        for name, kw in (("lone", dict(copies=1)), ("single", dict(copies=1, coop_off=True)), ("packed", dict(copies=257)), ("dense", dict(copies=1, dense=True))):
            try:
                Ws, b, s = gpu_run(w_hip, exc, **kw)
                k = len(Ws) - 1
                verdict, detail = compare(Ws[k], b, s[k], k, O, so, sens)
                lay = b.plan_stats()["layout"]
            except Exception as e:      # noqa: BLE001
                verdict, detail, lay = "ERROR", str(e)[:120], "?"
            tally[(name, verdict)] = tally.get((name, verdict), 0) + 1
            row.append(f"{name}:{verdict}" + (f"({detail})" if detail else "") + f"[{lay}]")
            if verdict in ("FIRST STEP", "STATE", "ERROR"):
                bad.append((seed, note, name, verdict, detail))
        np_, nl_, nlm = len(w_hip["proj"]["landmark"]), len(w_hip["line"]["frame"]), len(w_hip["lam"])
        print(f"case {seed} [{note}; {np_} point, {nl_} line factors, {nlm} landmarks, prior {'yes' if w_hip.get('prior') is not None else 'no'}]: " + "  ".join(row), flush=True)
    # the same windows six at a time in ONE batch (a lock-step frame: different structures side by side, the helpers sized by the largest
    # window, groups rotated over the XCDs), each against its own oracle solve
    for g0 in range(0, len(kept), 6):
        grp = kept[g0:g0 + 6]
        try:
            Ws = [tcv.Window(w, estimate_extrinsic=not exc) for (_, _, w, exc, _, _, _) in grp]
            b = tcv.Batch(Ws)
            b.solve(tcv.default_options(8, True, True, 256, True)); b.synchronize(); b.download_states()
            ss = b.summaries()
            for k, (seed, note, w, exc, O, so, sens) in enumerate(grp):
                verdict, detail = compare(Ws[k], b, ss[k], k, O, so, sens)
                tally[("mixed", verdict)] = tally.get(("mixed", verdict), 0) + 1
                if verdict in ("FIRST STEP", "STATE", "ERROR"):
                    bad.append((seed, note, "mixed", verdict, detail))
            print(f"mixed batch of seeds {[g[0] for g in grp]}: layout {b.plan_stats()['layout']}, plans {b.plan_stats()['num_plans']}, cooperative {b.cooperative()}", flush=True)
        except Exception as e:      # noqa: BLE001
            tally[("mixed", "ERROR")] = tally.get(("mixed", "ERROR"), 0) + 1
            bad.append((grp[0][0], "batch", "mixed", "ERROR", str(e)[:160]))
    print("\ntally:", {f"{a}/{b}": n for (a, b), n in sorted(tally.items())})
    print("mismatches:", len(bad))
    for x in bad:
        print("  ", x)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
