"""N4 on the CPU: the NumPy restatement of the 2D-3D line association against its committed known-answer
vectors and the properties the reference relies on (estimator.cpp:385-447, :602-885)."""
import numpy as np

import np_oracle as O
from util import load


def test_line_association_oracle_matches_golden_and_properties():
    z = load("lines.npz")
    W, H = int(z["width"]), int(z["height"])
    fov = np.array([O.lines_in_fov(z["poses"][k], z["ex"], z["Rbw"], z["Tbw"], z["K"], W, H, int(z["window_size"]), z["lines3d"]) for k in range(11)])
    assert np.array_equal(fov, z["in_fov"])
    hit = 0
    for q in range(0, len(z["det"]), 3):
        f = int(z["det_frame"][q])
        e, c, pv = O.line_correspondence_in_frame(z["poses"][f], z["ex"], z["Rbw"], z["Tbw"], z["K"], W, H, z["lines3d"], fov[f], z["det"][q],
                                                  float(z["angle_th"]), float(z["overlap_th"]))
        assert c == int(z["match"][q]) and np.array_equal(e, z["err"][q]) and np.array_equal(pv, z["proj"][q])
        if c >= 0:
            hit += 1
            assert fov[f][c] and 0 <= e[0] <= float(z["angle_th"]) and e[2] >= np.float32(z["overlap_th"]) - 1e-6 and 0 <= e[1] < 10000
        else:
            assert np.all(e == -1) and np.array_equal(pv, z["det"][q])       # the detected line itself is handed back (:703-712, :871-877)
    assert hit > 20


def test_line2d_point_to_segment_known_answers():
    l = O.Line2D([0.0, 0.0, 10.0, 0.0])
    assert np.allclose(l.point2flined(np.array([3.0, 4.0])), [3.0, 0.0])        # foot of the perpendicular inside the segment
    assert np.allclose(l.point2flined(np.array([-2.0, 1.0])), [0.0, 0.0])       # clamped to the nearer end point
    assert np.allclose(l.point2flined(np.array([14.0, -3.0])), [10.0, 0.0])
    a = O.Line2D([0.0, 0.0, 10.0, 0.0]); b = O.Line2D([2.0, 1.0, 8.0, 1.0])
    d, ov = O.cal_euler_dist(a, b)
    assert abs(d - 1.0) < 1e-12 and abs(ov - 0.6) < 1e-12                     # parallel at 1 px, 60 % overlap
    assert abs(O.cal_angle_dist(a, O.Line2D([0.0, 0.0, 1.0, 1.0])) - np.pi / 4) < 1e-12
