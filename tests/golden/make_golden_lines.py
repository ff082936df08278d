"""Generates tests/golden/lines.npz: known-answer vectors for the 2D-3D line association
(Estimator::UpdateLinesInFoV estimator.cpp:385-447, LineCorrespondenceInFrame :671-885) from the NumPy oracle.

    python tests/golden/make_golden_lines.py

11 frames of the synthetic trajectory, the 256-line subset of V1_01_easy/line_3d.txt as the prior map, and per frame:
noisy projections of fully visible map lines, of half-visible ones (the end-point walk-back branches :765-791 / :813-839),
rotated copies (fail the angle test), shifted short stubs (fail the overlap test) and random clutter (no match)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd"))
import np_oracle as O  # noqa: E402
import synth  # noqa: E402

W, H, WIN = 752, 480, 10
ANGLE_TH, OVERLAP_TH = 0.1745, 0.45      # sensor.yaml:119-122


def main():
    rng = np.random.default_rng(424242)
    ps, pe = synth.line_pool()
    lines_b = np.hstack([(ps - synth.TBW) @ synth.RBW, (pe - synth.TBW) @ synth.RBW])      # back into the prior-map frame
    ts = 2.0 + 0.35 * np.arange(11)
    poses = np.zeros((11, 7))
    for k, t in enumerate(ts):
        R = synth.traj_R(np.array([t]))[0]
        poses[k, :3] = synth.traj_p(np.array([t]))[0] + rng.normal(size=3) * 0.01
        q = O.R2q(R); poses[k, 3:] = q / np.linalg.norm(q) * (1 + 1e-9 * rng.normal())
    ex = np.concatenate([synth.TIC, O.R2q(synth.RIC)])
    fov = np.array([O.lines_in_fov(poses[k], ex, synth.RBW, synth.TBW, synth.K_MAT, W, H, WIN, lines_b) for k in range(11)])
    det_frame, det = [], []
    for k in range(11):
        R, T = O._line_extrinsic(poses[k], ex, synth.RBW, synth.TBW)
        pcs = lines_b[:, :3] @ R.T + T; pce = lines_b[:, 3:] @ R.T + T
        ok = (pcs[:, 2] > 0.1) & (pce[:, 2] > 0.1)
        us = synth.FX * pcs[:, 0] / np.where(ok, pcs[:, 2], 1) + synth.CX; vs = synth.FY * pcs[:, 1] / np.where(ok, pcs[:, 2], 1) + synth.CY
        ue = synth.FX * pce[:, 0] / np.where(ok, pce[:, 2], 1) + synth.CX; ve = synth.FY * pce[:, 1] / np.where(ok, pce[:, 2], 1) + synth.CY
        ins = ok & (us > 0) & (us < W - 1) & (vs > 0) & (vs < H - 1); ine = ok & (ue > 0) & (ue < W - 1) & (ve > 0) & (ve < H - 1)
        for j in np.nonzero(ins & ine)[0][:6]:
            v = np.array([us[j], vs[j], ue[j], ve[j]]) + rng.normal(size=4) * 1.0
            det_frame.append(k); det.append(v)
            c = 0.5 * (v[:2] + v[2:]); d = 0.5 * (v[2:] - v[:2]); a = 0.5      # rotated by 0.5 rad: fails angle_th
            Rm = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
            det_frame.append(k); det.append(np.concatenate([c - Rm @ d, c + Rm @ d]))
            det_frame.append(k); det.append(np.concatenate([v[2:] + 0.02 * (v[2:] - v[:2]), v[2:] + 0.3 * (v[2:] - v[:2])]) + 3.0)   # stub beyond the end
        for j in np.nonzero(ins ^ ine)[0][:4]:      # half-visible map lines: detected part = the visible half
            a = np.array([us[j], vs[j]]) if ins[j] else np.array([ue[j], ve[j]])
            b = np.array([ue[j], ve[j]]) if ins[j] else np.array([us[j], vs[j]])
            tt = 1.0
            while tt > 0 and not (0 < (a + tt * (b - a))[0] < W - 1 and 0 < (a + tt * (b - a))[1] < H - 1):
                tt -= 0.05
            if tt > 0.2:
                det_frame.append(k); det.append(np.concatenate([a, a + tt * (b - a)]) + rng.normal(size=4) * 0.5)
        for _ in range(3):
            p0 = np.array([rng.uniform(0, W), rng.uniform(0, H)]); det_frame.append(k); det.append(np.concatenate([p0, p0 + rng.normal(size=2) * 60]))
    det_frame = np.array(det_frame, dtype=np.int32); det = np.array(det)
    match, err, proj = [], [], []
    for f, v in zip(det_frame, det):
        e, c, pv = O.line_correspondence_in_frame(poses[f], ex, synth.RBW, synth.TBW, synth.K_MAT, W, H, lines_b, fov[f], v, ANGLE_TH, OVERLAP_TH)
        match.append(c); err.append(e); proj.append(pv)
    match = np.array(match, dtype=np.int32)
    print("detections", len(det), "matched", int((match >= 0).sum()), "in fov per frame", fov.sum(1))
    np.savez_compressed(os.path.join(HERE, "lines.npz"), poses=poses, ex=ex, Rbw=synth.RBW, Tbw=synth.TBW, K=synth.K_MAT, width=W, height=H,
                        window_size=WIN, angle_th=ANGLE_TH, overlap_th=OVERLAP_TH, lines3d=lines_b, det_frame=det_frame, det=det,
                        in_fov=fov, match=match, err=np.array(err, dtype=np.float32), proj=np.array(proj))


if __name__ == "__main__":
    main()
