"""N4 on the GPU: tcv_match_lines (UpdateLinesInFoV + LineCorrespondenceInFrame, estimator.cpp:385-447 / :671-885) against
the golden vectors.  Indices and FoV masks must be identical; the float-typed errors equal to a float ulp (device acos /
sqrt vs libm), the projected end points to 1e-9 px."""
import numpy as np
import pytest

from util import load

pytestmark = pytest.mark.gpu


def test_match_lines_golden(gpu):
    z = load("lines.npz")
    fov, match, err, proj = gpu.match_lines(z["poses"], z["ex"], z["Rbw"], z["Tbw"], z["K"], int(z["width"]), int(z["height"]), int(z["window_size"]),
                                            z["lines3d"], z["det_frame"], z["det"], float(z["angle_th"]), float(z["overlap_th"]))
    assert np.array_equal(fov, z["in_fov"])
    assert np.array_equal(match, z["match"])
    assert np.abs(err - z["err"]).max() <= 4e-6 * max(1.0, np.abs(z["err"]).max())
    assert np.abs(proj - z["proj"]).max() < 1e-9
    assert (match >= 0).sum() > 100 and (match < 0).sum() > 100


def test_match_lines_edge_cases(gpu):
    z = load("lines.npz")
    args = (z["ex"], z["Rbw"], z["Tbw"], z["K"], int(z["width"]), int(z["height"]), int(z["window_size"]))
    # no detections: only the FoV masks
    fov, match, err, proj = gpu.match_lines(z["poses"], *args, z["lines3d"], np.zeros(0, np.int32), np.zeros((0, 4)), 0.1745, 0.45)
    assert np.array_equal(fov, z["in_fov"]) and len(match) == 0
    # a map far behind the camera: nothing in the FoV, every detection comes back unmatched with the -1 error triple (:703-712)
    far = z["lines3d"] + 1e4
    fov, match, err, proj = gpu.match_lines(z["poses"][:1], *args, far, z["det_frame"][:5] * 0, z["det"][:5], 0.1745, 0.45)
    assert not fov.any() and np.all(match == -1) and np.all(err == -1) and np.array_equal(proj, z["det"][:5])
    # a single map line, a single frame
    j = int(z["match"][0]); f = int(z["det_frame"][0])
    fov, match, err, proj = gpu.match_lines(z["poses"][f:f + 1], *args, z["lines3d"][j:j + 1], [0], z["det"][:1], 0.1745, 0.45)
    assert match[0] == 0 and abs(err[0, 1] - z["err"][0, 1]) < 1e-5
    with pytest.raises(gpu.TcvError):
        gpu.match_lines(z["poses"], *args, z["lines3d"], [99], z["det"][:1], 0.1745, 0.45)


def test_fov_of_the_new_frame_and_matching_in_one_call(gpu):
    """fov_given = 2 + f (tcv.h): the FoV sets of the frames already in the window are frozen inputs, the row of frame f is computed in
    the same call and returned -- what the estimator does once per image instead of an FoV call followed by a matching call."""
    z = load("lines.npz")
    args = (z["ex"], z["Rbw"], z["Tbw"], z["K"], int(z["width"]), int(z["height"]), int(z["window_size"]))
    nf = z["poses"].shape[0]
    f = nf - 1
    frozen = np.array(z["in_fov"], dtype=np.uint8)
    rng = np.random.default_rng(5)
    frozen[:f] ^= (rng.random(frozen[:f].shape) < 0.02).astype(np.uint8)      # frozen sets need not equal what the current poses would give
    two = frozen.copy()
    fov_now, _, _, _ = gpu.match_lines(z["poses"], *args, z["lines3d"], np.zeros(0, np.int32), np.zeros((0, 4)), float(z["angle_th"]), float(z["overlap_th"]))
    two[f] = fov_now[f]
    _, m2, e2, p2 = gpu.match_lines(z["poses"], *args, z["lines3d"], z["det_frame"], z["det"], float(z["angle_th"]), float(z["overlap_th"]), in_fov=two)
    one = frozen.copy(); one[f] = 255 - one[f] * 0      # garbage in the row that is to be computed
    fov1, m1, e1, p1 = gpu.match_lines(z["poses"], *args, z["lines3d"], z["det_frame"], z["det"], float(z["angle_th"]), float(z["overlap_th"]), in_fov=one, fov_frame=f)
    assert np.array_equal(fov1, two.astype(bool)) and np.array_equal(m1, m2) and np.array_equal(e1, e2) and np.array_equal(p1, p2)
    with pytest.raises(gpu.TcvError):
        gpu.match_lines(z["poses"], *args, z["lines3d"], z["det_frame"], z["det"], 0.1745, 0.45, in_fov=one, fov_frame=nf)


def test_batched_association_equals_the_single_calls(gpu):
    """tcv_match_lines_batch: the associations of the sequences of a lock-step frame -- each with its own map, poses, detections and FoV
    mode -- in ONE device round trip; every call gives exactly what tcv_match_lines gives alone"""
    z = load("lines.npz")
    base = dict(ex_pose=z["ex"], Rbw=z["Rbw"], Tbw=z["Tbw"], K=z["K"], width=int(z["width"]), height=int(z["height"]), window_size=int(z["window_size"]),
                angle_th=float(z["angle_th"]), overlap_th=float(z["overlap_th"]))
    nf = z["poses"].shape[0]
    frozen = np.array(z["in_fov"], dtype=np.uint8)
    calls = [dict(base, poses=z["poses"], lines3d=z["lines3d"], det_frame=z["det_frame"], det_lines=z["det"]),                                        # FoV computed
             dict(base, poses=z["poses"], lines3d=z["lines3d"][::2], det_frame=z["det_frame"][:40], det_lines=z["det"][:40]),                          # another map
             dict(base, poses=z["poses"], lines3d=z["lines3d"], det_frame=z["det_frame"], det_lines=z["det"], in_fov=frozen, fov_frame=nf - 1),        # steady state
             dict(base, poses=z["poses"][:3], lines3d=z["lines3d"], det_frame=np.zeros(0, np.int32), det_lines=np.zeros((0, 4))),                      # no detections
             dict(base, poses=z["poses"], lines3d=z["lines3d"], det_frame=z["det_frame"][5:9], det_lines=z["det"][5:9], in_fov=frozen)]                # every row given
    got = gpu.match_lines_batch(calls)
    for c, g in zip(calls, got):
        ref = gpu.match_lines(c["poses"], c["ex_pose"], c["Rbw"], c["Tbw"], c["K"], c["width"], c["height"], c["window_size"], c["lines3d"], c["det_frame"], c["det_lines"],
                              c["angle_th"], c["overlap_th"], in_fov=c.get("in_fov"), fov_frame=c.get("fov_frame"))
        for a, b in zip(g, ref):
            assert np.array_equal(a, b)
    assert np.array_equal(got[0][1], z["match"])
