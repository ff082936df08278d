"""Shared helpers for the tests: golden-vector loading and small comparison utilities."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
sys.path.insert(0, GOLDEN)


def rel(a, b):
    a = np.asarray(a, dtype=float); b = np.asarray(b, dtype=float)
    if a.size == 0 and b.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def fro(a, b):
    a = np.asarray(a, dtype=float); b = np.asarray(b, dtype=float)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_windows():
    """(pre window, main window with the golden prior, npz) from tests/golden/window.npz."""
    from make_golden import unflatten_window
    z = load("window.npz")
    pre = unflatten_window(z, "pre_"); main = unflatten_window(z, "main_")
    kinds = {0: "pose", 1: "sb", 2: "ex"}
    sizes = [int(s) for s in z["marg_sizes"]]
    x0, o = [], 0
    for s in sizes:
        x0.append(z["marg_x0"][o:o + s].copy()); o += s
    main["prior"] = dict(m=int(z["marg_m"]), n=int(z["marg_n"]), sizes=sizes, idx=[int(i) for i in z["marg_idx"]], x0=x0,
                         J0=z["marg_J0"], r0=z["marg_r0"],
                         blocks=[(kinds[int(k)], int(i)) for k, i in zip(z["marg_block_kind"], z["marg_block_index"])])
    return pre, main, z


def imu_pre(z, k, prefix="i1_"):
    return dict(delta_p=z[prefix + "delta_p"][k], delta_q=z[prefix + "delta_q"][k], delta_v=z[prefix + "delta_v"][k],
                lin_ba=z[prefix + "lin_ba"][k], lin_bg=z[prefix + "lin_bg"][k], sum_dt=float(z[prefix + "sum_dt"][k]),
                jacobian=z[prefix + "jacobian"][k], covariance=z[prefix + "covariance"][k])


def sub_window(w, frames, keep_lines=True):
    """first `frames` frames of a window (ragged case: factors that touch later frames are dropped)."""
    out = dict(w)
    out["pose"] = w["pose"][:frames]; out["speedbias"] = w["speedbias"][:frames]
    im = w["imu"]; ki = [k for k in range(len(im["frame_i"])) if im["frame_j"][k] < frames]
    out["imu"] = {k: (np.asarray(v)[ki] if isinstance(v, np.ndarray) and v.shape[:1] == (len(im["frame_i"]),) else v) for k, v in im.items()}
    pr = w["proj"]; kp = [k for k in range(len(pr["frame_i"])) if pr["frame_j"][k] < frames]
    out["proj"] = {k: (np.asarray(v)[kp] if isinstance(v, np.ndarray) and v.shape[:1] == (len(pr["frame_i"]),) else v) for k, v in pr.items()}
    used = sorted(set(int(l) for l in out["proj"]["landmark"]))
    remap = {l: i for i, l in enumerate(used)}
    out["proj"]["landmark"] = np.array([remap[int(l)] for l in out["proj"]["landmark"]], int)
    out["lam"] = w["lam"][used]
    ln = w["line"]; kl = [k for k in range(len(ln["frame"])) if ln["frame"][k] < frames and keep_lines]
    out["line"] = {k: (np.asarray(v)[kl] if isinstance(v, np.ndarray) and v.shape[:1] == (len(ln["frame"]),) else v) for k, v in ln.items()}
    out["prior"] = None
    return out
