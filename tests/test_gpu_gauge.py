"""G3 on the GPU: tcv_gauge_fix / tcv_batch_gauge_fix against the golden vectors and the C oracle
(Estimator::double2vector, estimator.cpp:1537-1581).  Pure FP64 arithmetic with libm-level trigonometry: 1e-12."""
import numpy as np
import pytest

import np_oracle as O
import orc
import synth
from util import load, rel

pytestmark = pytest.mark.gpu


def test_gauge_fix_golden(gpu):
    z = load("gauge.npz")
    for k in range(len(z["kinds"])):
        Rs, Ps, Vs, po = gpu.gauge_fix(z["R0"][k], z["P0"][k], z["pose"][k], z["sb"][k])
        assert rel(Rs, z["Rs"][k]) < 1e-12 and rel(Ps, z["Ps"][k]) < 1e-12 and rel(Vs, z["Vs"][k]) < 1e-12, z["kinds"][k]
        assert rel(po, z["pose_out"][k]) < 1e-11, z["kinds"][k]


def test_gauge_fix_edge_cases(gpu):
    z = load("gauge.npz")
    Rs, Ps, Vs, po = gpu.gauge_fix(z["R0"][0], z["P0"][0], z["pose"][0][:1], z["sb"][0][:1])       # a single frame
    assert np.allclose(Ps[0], z["P0"][0]) and Rs.shape == (1, 3, 3)
    bad = z["pose"][0].copy(); bad[3, 4] = np.nan
    with pytest.raises(gpu.TcvError):
        gpu.gauge_fix(z["R0"][0], z["P0"][0], bad, z["sb"][0])


def test_batch_gauge_fix_in_place_after_solve(gpu):
    """solve -> gauge fix in HBM -> download: equals the oracle's double2vector applied to the oracle-solved states, and the
    marginalisation that follows linearises at the gauge-fixed states (estimator.cpp:1905, :1915)."""
    batch = synth.make_windows(4242, 3, frame_shift=-1)
    wins = [synth.window_at(batch, k) for k in range(3)]
    W = [gpu.Window(w) for w in wins]
    MW = [gpu.margin_old_window(w) for w in wins]
    M = [gpu.Window(mw, share=W[k]) for k, mw in enumerate(MW)]
    drops = [gpu.margin_old_drops(W[k], MW[k]) for k in range(3)]
    b = gpu.Batch(W, M, drops)
    b.solve(gpu.default_options(8, True)); b.gauge_fix(); b.marginalize(); b.synchronize(); b.download_states()
    for k in range(3):
        Oc = orc.Window(wins[k]); Oc.solve(8, True); st = Oc.states()
        R0 = O.q2R(np.asarray(wins[k]["pose"])[0, 3:]); P0 = np.asarray(wins[k]["pose"])[0, :3]
        Rs, Ps, Vs, po = orc.gauge_fix(R0, P0, st["pose"], st["sb"])
        assert rel(W[k].pose[:, :3], Ps) < 1e-6 and rel(W[k].sb[:, :3], Vs) < 1e-6
        for i in range(po.shape[0]):
            assert np.abs(O.q2R(W[k].pose[i, 3:]) - Rs[i]).max() < 1e-6
        assert np.allclose(W[k].pose[0, :3], P0, atol=1e-13)
        assert rel(W[k].sb[:, 3:], st["sb"][:, 3:]) < 1e-6             # biases are not touched by the gauge fix
        # ... and equals the oracle's marginalisation at the oracle's gauge-fixed states, within the reproducibility floor
        st2 = dict(st); sbf = st["sb"].copy(); sbf[:, :3] = Vs
        w2 = dict(wins[k], pose=po, speedbias=sbf, ex_pose=st["ex"], lam=st["lam"])
        pref, dbg = orc.Window(w2).marginalize_old()
        As, bs = b.prior(k).schur()
        assert np.linalg.norm(As - dbg["A_schur"]) < 1e-5 * np.linalg.norm(dbg["A_schur"])
        assert np.linalg.norm(bs - dbg["b_schur"]) < 1e-5 * np.linalg.norm(dbg["b_schur"])
        # the new prior's linearisation point is the gauge-fixed state (preMarginalize after vector2double)
        d = b.prior(k).export()
        kept = gpu.shifted_prior_blocks(b.prior(k), W[k])
        for (name, idx), x0 in zip(kept, d["x0"]):
            cur = {"pose": W[k].pose, "sb": W[k].sb}.get(name)
            if cur is not None:
                assert np.array_equal(x0, cur[idx + 1]), (name, idx)


def test_gauge_fix_in_the_solve_kernels_epilogue_is_the_stand_alone_kernel_bit_for_bit(gpu):
    """tcv_batch_set_fused_gauge_fix: double2vector() applied to the solved states while they are still in LDS (one launch less per frame).  Same
    arithmetic as tcv_batch_gauge_fix's kernel -- states, the marginalisation that linearises at them, and a later stand-alone call (a no-op) all
    agree bit for bit; in the packed two-per-CU shape, one workgroup per window, the cooperative mode and the dense layout."""
    L = gpu.lib()

    def run(B, seed, fused, variant=0, wg=0):
        L.tcv_set_solver_variant(variant)
        try:
            batch = synth.make_windows(seed, B, frame_shift=-1)
            wins = [synth.window_at(batch, k) for k in range(B)]
            W = [gpu.Window(w) for w in wins]
            MW = [gpu.margin_old_window(w) for w in wins]
            M = [gpu.Window(mw, share=W[k]) for k, mw in enumerate(MW)]
            drops = [gpu.margin_old_drops(W[k], MW[k]) for k in range(B)]
            b = gpu.Batch(W, M, drops)
            if fused:
                b.fuse_gauge_fix()
            o = gpu.default_options(8, True)
            o.workgroups_per_window = wg
            b.solve(o); b.gauge_fix(); b.marginalize(); b.synchronize(); b.download_states()
            pri = [b.prior(k).export() for k in range(B)]
            return [(W[k].pose.copy(), W[k].sb.copy(), W[k].ex.copy(), W[k].lam.copy(), pri[k]["J0"], pri[k]["r0"]) for k in range(B)]
        finally:
            L.tcv_set_solver_variant(0)

    for B, seed, variant, wg in ((3, 4300, 0, 0), (3, 4300, 0, 1), (300, 4400, 0, 0), (3, 4300, 1, 0)):      # cooperative | one workgroup | two per CU | dense
        a, f = run(B, seed, False, variant, wg), run(B, seed, True, variant, wg)
        for x, y in zip(a, f):
            for u, v in zip(x, y):
                assert np.array_equal(u, v)
