"""Builds tc-viml_amd/data/euroc_<seq>.npz: excerpts of the DATA files the reference ships next to its per-sequence
configuration (benchmark_publisher/config/<seq>/): the EuRoC ground-truth states `data.csv` (17 columns at 200 Hz, the rows
benchmark_publisher_node.cpp:42-62 parses), the prior 3D line map `line_3d.txt` (n x 6, map frame) and, from `sensor.yaml`,
the numbers the estimator reads: initialRotation / initialTranslation (map -> VIO world, :40-57), the camera intrinsics and
the camera-IMU extrinsics.  V1_01's ground truth is one of the reference's missing blobs, so the five sequences that have
one are used.

Data fixture only (no reference source text); run in the authoring container, /root/reference does not exist on the GPU
box.  The CSV carries 6 decimals, so positions, quaternions, velocities and biases are stored exactly as integer
micro-units, delta-coded along time (what makes the archive small).
"""
import os
import re

import numpy as np

CFG = "/root/reference/benchmark_publisher/config"
SEQS = ["V1_02_medium", "V1_03_difficult", "V2_01_easy", "V2_02_medium", "V2_03_difficult"]
START_S, LENGTH_S = 4.0, 36.0          # skip the hand-held start-up wiggle the estimator initialises on


def yaml_matrix(text, name, count):
    """the last un-commented `data: [...]` after `name:`."""
    body = text[text.index(name + ":"):]
    body = body[:body.index("]") + 4096]
    lines = [l for l in body.split("\n") if not l.strip().startswith("#")]
    m = re.search(r"data:\s*\[([^\]]*)\]", "\n".join(lines))
    vals = [float(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()]
    assert len(vals) == count, (name, vals)
    return np.array(vals)


def yaml_scalar(text, name):
    for l in text.split("\n"):
        if l.strip().startswith(name + ":"):
            return float(l.split(":")[1].split("#")[0])
    raise KeyError(name)


def main():
    out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tc-viml_amd", "data")
    for seq in SEQS:
        gt = np.loadtxt(os.path.join(CFG, seq, "data.csv"), delimiter=",", skiprows=1, usecols=range(17), dtype=np.float64)
        # the stamps are nanoseconds beyond 2^53: read them as integers
        with open(os.path.join(CFG, seq, "data.csv")) as f:
            next(f)
            stamps = np.array([int(l.split(",")[0]) for l in f if l.strip()], dtype=np.int64)
        t = (stamps - stamps[0]) * 1e-9
        sel = np.nonzero((t >= START_S) & (t < START_S + LENGTH_S))[0]
        micro = np.rint(gt[sel, 1:] * 1e6).astype(np.int64)
        assert np.abs(micro * 1e-6 - gt[sel, 1:]).max() < 1e-9
        delta = np.diff(micro, axis=0, prepend=0).astype(np.int32)
        assert (np.cumsum(delta.astype(np.int64), axis=0) == micro).all()
        lines = np.loadtxt(os.path.join(CFG, seq, "line_3d.txt"))
        y = open(os.path.join(CFG, seq, "sensor.yaml")).read()
        np.savez_compressed(
            os.path.join(out_dir, "euroc_%s.npz" % seq),
            stamp_ns=stamps[sel], state_micro_delta=delta,          # columns: p(3) q(wxyz) v(3) bw(3) ba(3)
            lines3d=lines,
            Rbw=yaml_matrix(y, "initialRotation", 9).reshape(3, 3), Tbw=yaml_matrix(y, "initialTranslation", 3),
            Ric=yaml_matrix(y, "extrinsicRotation", 9).reshape(3, 3), Tic=yaml_matrix(y, "extrinsicTranslation", 3),
            K=np.array([yaml_scalar(y, "fx"), yaml_scalar(y, "fy"), yaml_scalar(y, "cx"), yaml_scalar(y, "cy")]),
            size=np.array([yaml_scalar(y, "width"), yaml_scalar(y, "height")]),
            imu_noise=np.array([yaml_scalar(y, k) for k in ("acc_n", "gyr_n", "acc_w", "gyr_w", "g_norm")]),
            line_th=np.array([yaml_scalar(y, k) for k in ("angle_th", "overlap_th", "dist_th")]))      # estimator.cpp:116-119
        p = os.path.join(out_dir, "euroc_%s.npz" % seq)
        print(seq, "rows", len(sel), "lines", lines.shape, "bytes", os.path.getsize(p))


if __name__ == "__main__":
    main()
