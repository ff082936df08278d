"""Relocalisation factors (estimator.cpp:1854-1886): `relo_Pose` as a 12th pose block with PoseLocalParameterization and one more
ProjectionFactor per matched landmark on (para_Pose[start], relo_Pose, para_Ex_Pose[0], para_Feature[idx]).  The camera side of the
window grows to 177 tangent dims (12 x 6 + 6 + 11 x 9): the chain layout holds it (pose system 78 + 1 wide, still five tile rows; the
vectors over the camera tangent space are 184 wide for such a plan, PlanHdr::camw).  HIP path vs the NumPy oracle."""
import ctypes as C

import numpy as np
import pytest

import np_oracle as NO
import synth
from relo_util import add_relocalisation
from util import golden_windows, rel

pytestmark = pytest.mark.gpu


def hip_relo_window(tcv, w):
    W = tcv.Window(w)
    rl = w["relo"]
    relo = tcv.f64(rl["pose"]).copy()
    L = tcv.lib()
    tcv.check(L.tcv_problem_add_parameter_block(W.h, tcv.dptr(relo), 7, tcv.TCV_PARAM_POSE))       # :1857-1858
    pr = w["proj"]
    keep = []
    for k in range(len(rl["frame_i"])):
        pi, pj = tcv.f64(rl["pts_i"][k]).copy(), tcv.f64(rl["pts_j"][k]).copy()
        keep.append((pi, pj))
        tcv.check(L.tcv_problem_add_projection_factor(W.h, tcv.dptr(pi), tcv.dptr(pj), float(pr["sqrt_info"]), float(pr["loss_a"]),
                                                      W.block_ptr("pose", int(rl["frame_i"][k])), tcv.dptr(relo), tcv.dptr(W.ex), W.block_ptr("lam", int(rl["landmark"][k]))))      # :1878-1880
    W._relo_keep = keep
    return W, relo


@pytest.mark.parametrize("which", ["golden_with_prior_and_lines", "synthetic_points_only"])
def test_relocalisation_window_vs_oracle(gpu, which):
    if which == "golden_with_prior_and_lines":
        pre, main, z = golden_windows()
        w = add_relocalisation(main, f=4, seed=1)
    else:
        w = add_relocalisation(dict(synth.window_at(synth.make_windows(9300, 1, with_lines=False), 0), prior=None), f=6, seed=2)
    assert len(w["relo"]["frame_i"]) >= 8
    W, relo = hip_relo_window(gpu, w)
    st = W.plan_stats()
    assert st["nc"] == 177 and st["npp"] == 78 and st["lds_bytes"] <= 160 * 1024
    assert gpu.lib().tcv_problem_num_parameter_blocks(W.h) == 11 + 11 + 1 + 1 + len(w["lam"])
    b = gpu.Batch([W])
    assert b.plan_stats()["layout"] == "chain"
    for iters, fixed in ((8, True), (30, False)):
        W, relo = hip_relo_window(gpu, w)
        b = gpu.Batch([W])
        b.solve(gpu.default_options(iters, fixed)); b.synchronize(); b.download_states()
        s = b.summaries()[0]
        x, so = NO.solve(NO.Problem(w), iters, fixed)
        its = so["iterations"]
        assert s.num_iterations == len(its)
        assert [bool(s.step_ok[i]) for i in range(1, len(its))] == [bool(r.get("step_ok", False)) for r in its[1:]]
        assert [s.dogleg_case[i] for i in range(1, len(its))] == [int(r.get("dogleg_case", r.get("case", -1))) for r in its[1:]]
        assert abs(s.final_cost - so["final_cost"]) < 1e-6 * so["final_cost"]
        assert s.final_cost < s.initial_cost
        for key, got in (("pose", W.pose), ("sb", W.sb), ("ex", W.ex), ("lam", W.lam), ("relo", relo)):
            assert rel(got, x[key]) < 1e-6, key
        assert np.abs(relo - w["relo"]["pose"]).max() > 1e-4          # the relocalisation pose is a free block: it moved


def test_relocalisation_window_in_a_large_batch(gpu):
    """more windows than CUs (two workgroups per CU, 80 KiB each): relocalisation windows next to plain ones, each giving what it gives alone"""
    B = 300
    batch = synth.make_windows(9400, B)
    wins = [synth.window_at(batch, k) for k in range(B)]
    Ws, relos = [], {}
    for k in range(B):
        if k % 5 == 0:
            w = add_relocalisation(dict(wins[k], prior=None), f=3 + (k // 5) % 5, seed=k)
            W, relo = hip_relo_window(gpu, w); relos[k] = (w, relo)
        else:
            W = gpu.Window(wins[k])
        Ws.append(W)
    b = gpu.Batch(Ws)
    assert b.plan_stats()["lds_bytes"] == 80 * 1024
    b.solve(gpu.default_options(8, True)); b.synchronize(); b.download_states()
    s = b.summaries()
    for k in (0, 5, 295):
        w, relo = relos[k]
        W1, relo1 = hip_relo_window(gpu, w)
        b1 = gpu.Batch([W1]); b1.solve(gpu.default_options(8, True, workgroups_per_window=1)); b1.synchronize(); b1.download_states()
        s1 = b1.summaries()[0]
        assert abs(s[k].final_cost - s1.final_cost) < 1e-9 * s1.final_cost and rel(relo, relo1) < 1e-9 and rel(Ws[k].pose, W1.pose) < 1e-9
    assert all(np.isfinite(s[k].final_cost) and s[k].final_cost < s[k].initial_cost for k in range(B))
