"""Opt-in extension tcv_problem_set_line_jacobian(p, 1) (NOT the reference's behaviour; default off): the line factors use the
derivative of their residual instead of LineProjectionFactor's Jacobian as written.  The kernel against the NumPy restatement, and
what it does to a replay whose line observations all carry their true 3D partner."""
import numpy as np
import pytest

import ate
import np_oracle as NO
import replay
import synth
from util import rel

pytestmark = pytest.mark.gpu


def test_exact_line_jacobian_solve_vs_oracle(gpu):
    w = synth.window_at(synth.make_windows(77, 1), 0)
    w = dict(w, line=dict(w["line"], exact_jacobian=True))
    W = gpu.Window(w); b = gpu.Batch([W])
    b.solve(gpu.default_options(8, True)); b.synchronize(); b.download_states()
    s = b.summaries()[0]
    x, so = NO.solve(NO.Problem(w), 8, True)
    assert abs(s.final_cost - so["final_cost"]) < 1e-6 * so["final_cost"]
    assert [s.step_ok[i] for i in range(1, 9)] == [int(it["step_ok"]) for it in so["iterations"][1:]]
    assert rel(W.pose, x["pose"]) < 1e-6 and rel(W.sb, x["sb"]) < 1e-6
    # and it is a different system from the default one
    w0 = dict(w, line=dict(w["line"], exact_jacobian=False))
    W0 = gpu.Window(w0); b0 = gpu.Batch([W0]); b0.solve(gpu.default_options(8, True)); b0.synchronize(); b0.download_states()
    assert rel(W0.pose, W.pose) > 1e-6


def test_exact_line_jacobian_turns_the_prior_map_into_an_anchor(gpu):
    """V1_03_difficult, 19 s, every line observation with its true 3D partner: with the reference's Jacobian the line terms push the
    estimate away from the map, with the exact one they hold it in place -- better than without any line factor."""
    kw = dict(start_s=0.5, max_features=40)
    st_lines = replay.simulate_stream_euroc("V1_03_difficult", 200, max_lines=8, **kw)
    st_none = replay.simulate_stream_euroc("V1_03_difficult", 200, max_lines=0, **kw)
    ref = replay.run_many_native([st_lines], num_iterations=8)[0]
    exact = replay.run_many_native([st_lines], num_iterations=8, exact_line_jacobian=True)[0]
    none = replay.run_many_native([st_none], num_iterations=8)[0]
    gt = st_lines["gt_p"][replay.WINDOW_SIZE:]
    e_ref, e_exact, e_none = (ate.ate_rmse(o["p"], gt) for o in (ref, exact, none))
    print("SE(3)-aligned ATE vs ground truth [m]: reference Jacobian %.4f, exact %.4f, no lines %.4f" % (e_ref, e_exact, e_none))
    assert e_exact < 0.02 and e_exact < 0.8 * e_none and e_ref > 3.0 * e_exact
