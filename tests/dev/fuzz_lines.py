#!/usr/bin/env python3
"""Developer checker (GPU): the 2D-3D line association (UpdateLinesInFoV + LineCorrespondenceInFrame, estimator.cpp:385-447 / :671-885) on
seeded random scenes against the NumPy oracle -- the generator of tests/golden/make_golden_lines.py with its seed, start time, pose noise,
detection noise and map subset drawn per scene.  Field-of-view masks and match indices must be IDENTICAL (index work); the float-typed errors
within a float ulp, the projected end points within 1e-9 px.  A detection whose decision sits on a threshold in float arithmetic may
legitimately flip between libm and the device's acos / sqrt: such cases are listed with the oracle's margins.

    python tests/dev/fuzz_lines.py [scenes] [first seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import np_oracle as O      # noqa: E402
import synth      # noqa: E402
import tcv      # noqa: E402

W, H, WIN = 752, 480, 10
ANGLE_TH, OVERLAP_TH = 0.1745, 0.45


def scene(seed):
    rng = np.random.default_rng(900000 + seed)
    ps, pe = synth.line_pool()
    lines_b = np.hstack([(ps - synth.TBW) @ synth.RBW, (pe - synth.TBW) @ synth.RBW])
    sub = rng.permutation(len(lines_b))[: int(rng.integers(1, len(lines_b) + 1))]
    lines_b = lines_b[np.sort(sub)]
    nf = int(rng.integers(1, 12))
    ts = rng.uniform(0.5, 30.0) + 0.35 * np.arange(nf)
    pn = float(rng.choice([0.0, 0.01, 0.2]))
    poses = np.zeros((nf, 7))
    for k, t in enumerate(ts):
        R = synth.traj_R(np.array([t]))[0]
        poses[k, :3] = synth.traj_p(np.array([t]))[0] + rng.normal(size=3) * pn
        q = O.R2q(R); poses[k, 3:] = q / np.linalg.norm(q) * (1 + 1e-9 * rng.normal())
    ex = np.concatenate([synth.TIC, O.R2q(synth.RIC)])
    fov = np.array([O.lines_in_fov(poses[k], ex, synth.RBW, synth.TBW, synth.K_MAT, W, H, WIN, lines_b) for k in range(nf)])
    dn = float(rng.choice([0.0, 0.3, 1.0, 3.0]))
    det_frame, det = [], []
    for k in range(nf):
        R, T = O._line_extrinsic(poses[k], ex, synth.RBW, synth.TBW)
        pcs = lines_b[:, :3] @ R.T + T; pce = lines_b[:, 3:] @ R.T + T
        ok = (pcs[:, 2] > 0.1) & (pce[:, 2] > 0.1)
        z0 = np.where(ok, pcs[:, 2], 1); z1 = np.where(ok, pce[:, 2], 1)
        us = synth.FX * pcs[:, 0] / z0 + synth.CX; vs = synth.FY * pcs[:, 1] / z0 + synth.CY
        ue = synth.FX * pce[:, 0] / z1 + synth.CX; ve = synth.FY * pce[:, 1] / z1 + synth.CY
        ins = ok & (us > 0) & (us < W - 1) & (vs > 0) & (vs < H - 1); ine = ok & (ue > 0) & (ue < W - 1) & (ve > 0) & (ve < H - 1)
        vis = np.nonzero(ins & ine)[0]
        for j in rng.permutation(vis)[:8]:
            v = np.array([us[j], vs[j], ue[j], ve[j]]) + rng.normal(size=4) * dn
            if rng.random() < 0.3:
                v = np.concatenate([v[2:], v[:2]])      # end points the other way round
            det_frame.append(k); det.append(v)
            c = 0.5 * (v[:2] + v[2:]); d = 0.5 * (v[2:] - v[:2]); a = float(rng.uniform(0.05, 0.6))      # rotated: around the angle threshold
            Rm = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
            det_frame.append(k); det.append(np.concatenate([c - Rm @ d, c + Rm @ d]))
            s0, s1 = float(rng.uniform(-0.5, 0.9)), float(rng.uniform(0.1, 1.5))      # a piece along the line: around the overlap threshold
            det_frame.append(k); det.append(np.concatenate([v[:2] + s0 * (v[2:] - v[:2]), v[:2] + s1 * (v[2:] - v[:2])]) + rng.normal(size=4) * 0.5)
        for j in np.nonzero(ins ^ ine)[0][:4]:
            a = np.array([us[j], vs[j]]) if ins[j] else np.array([ue[j], ve[j]])
            b = np.array([ue[j], ve[j]]) if ins[j] else np.array([us[j], vs[j]])
            tt = 1.0
            while tt > 0 and not (0 < (a + tt * (b - a))[0] < W - 1 and 0 < (a + tt * (b - a))[1] < H - 1):
                tt -= 0.05
            if tt > 0.2:
                det_frame.append(k); det.append(np.concatenate([a, a + tt * (b - a)]) + rng.normal(size=4) * 0.5)
        for _ in range(3):
            p0 = np.array([rng.uniform(0, W), rng.uniform(0, H)]); det_frame.append(k); det.append(np.concatenate([p0, p0 + rng.normal(size=2) * 60]))
    det_frame = np.array(det_frame, dtype=np.int32); det = np.array(det).reshape(-1, 4)
    match, err, proj = [], [], []
    for f, v in zip(det_frame, det):
        e, c, pv = O.line_correspondence_in_frame(poses[f], ex, synth.RBW, synth.TBW, synth.K_MAT, W, H, lines_b, fov[f], v, ANGLE_TH, OVERLAP_TH)
        match.append(c); err.append(e); proj.append(pv)
    return dict(poses=poses, ex=ex, lines=lines_b, det_frame=det_frame, det=det, fov=fov, match=np.array(match, dtype=np.int32),
                err=np.array(err, dtype=np.float32).reshape(-1, 3), proj=np.array(proj).reshape(-1, 4), note=f"{nf} frames, {len(lines_b)} map lines, pose noise {pn}, detection noise {dn}")


def main():
    scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    tot = dict(det=0, matched=0, fov_bits=0, fov_diff=0, match_diff=0)
    worst_err = worst_proj = 0.0
    bad = []
    for s in range(scenes):
        seed = seed0 + s
        z = scene(seed)
        fov, match, err, proj = tcv.match_lines(z["poses"], z["ex"], synth.RBW, synth.TBW, synth.K_MAT, W, H, WIN, z["lines"], z["det_frame"], z["det"], ANGLE_TH, OVERLAP_TH)
        nd = len(z["det"])
        fd = int(np.sum(np.asarray(fov, bool) != np.asarray(z["fov"], bool)))
        md = np.nonzero(np.asarray(match) != z["match"])[0] if nd else np.zeros(0, int)
        same = np.asarray(match) == z["match"] if nd else np.zeros(0, bool)
        if nd:
            ee = np.abs(np.asarray(err) - z["err"])[same]
            worst_err = max(worst_err, float((ee / np.maximum(1.0, np.abs(z["err"][same]))).max()) if ee.size else 0.0)
            pp = np.abs(np.asarray(proj) - z["proj"])[same]
            worst_proj = max(worst_proj, float(pp.max()) if pp.size else 0.0)
        tot["det"] += nd; tot["matched"] += int((z["match"] >= 0).sum()); tot["fov_bits"] += z["fov"].size; tot["fov_diff"] += fd; tot["match_diff"] += len(md)
        line = f"scene {seed} [{z['note']}]: {nd} detections, {int((z['match'] >= 0).sum())} matched; FoV bits differing {fd}, matches differing {len(md)}"
        for i in md[:5]:
            line += f"\n     detection {int(i)} (frame {int(z['det_frame'][i])}): device {int(match[i])} err {np.asarray(err)[i]}, oracle {int(z['match'][i])} err {z['err'][i]}"
            bad.append((seed, int(i)))
        print(line, flush=True)
    print("\ntotal:", tot, "worst relative error of the float-typed errors %.2e, worst projected end point difference %.2e px" % (worst_err, worst_proj))
    print("flagged detections:", len(bad), bad[:20])
    return 1 if (bad or tot["fov_diff"]) else 0


if __name__ == "__main__":
    sys.exit(main())
