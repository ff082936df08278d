"""Device-resident frame-to-frame state (include/tcv.h tcv_batch_get_priors_device): what `last_marginalization_info` is between two frames
of the reference (estimator.h:176-177, estimator.cpp:2027-2044).  The prior a batch's marginalisation produced stays in HBM; the next
batch's tcv_batch_create copies J0 | r0 | x0 device-to-device into its pool (without the thresholded rows, the layout of the host path).
Everything here is gated BIT FOR BIT against the host round trip (download, tcv_prior object, pack, upload) of the same chain."""
import os

import numpy as np
import pytest

import replay
import synth

pytestmark = pytest.mark.gpu


def _marg_batch(tcv, wins, Ws):
    MW = [tcv.margin_old_window(W.win) for W in Ws]      # (the dicts the windows were built from: they carry the prior's block list)
    M = [tcv.Window(mw, share=Ws[k], prior=Ws[k].prior) for k, mw in enumerate(MW)]
    drops = [tcv.margin_old_drops(Ws[k], MW[k]) for k in range(len(wins))]
    return tcv.Batch(Ws, M, drops)


def _run(b, tcv):
    b.solve(tcv.default_options(8, True)); b.gauge_fix(); b.marginalize(); b.synchronize(); b.download_states()
    return b.summaries()


def _same_summary(a, c):
    assert a.num_iterations == c.num_iterations and a.termination == c.termination and a.final_cost == c.final_cost
    m = min(a.num_iterations, 64)
    for f in ("cost", "cost_candidate", "model_cost_change", "radius", "rho", "step_norm"):
        assert [getattr(a, f)[i] for i in range(m)] == [getattr(c, f)[i] for i in range(m)], f


@pytest.mark.parametrize("B", [5, 300])      # cooperative small-batch shape / two workgroups per CU
def test_device_resident_prior_chain_is_bit_identical_to_the_host_round_trip(gpu, B):
    tcv = gpu
    pre = synth.make_windows(8800, B, frame_shift=-1)
    pw = [synth.window_at(pre, k) for k in range(B)]
    main = synth.make_windows(8800, B)
    mw = [synth.window_at(main, k) for k in range(B)]
    nw = mw      # (the generator knows two consecutive windows of a trajectory; as third link the main windows once more, on the prior their
                 # own marginalisation made: structurally a frame-to-frame hand-over like any other, which is what is compared bit for bit)

    def chain(device):
        """three chained solves: no prior -> prior of the first -> prior of the second"""
        W0 = [tcv.Window(w) for w in pw]
        b0 = _marg_batch(tcv, pw, W0)
        _run(b0, tcv)
        out = []
        prev_b, prev_W = b0, W0
        for frame in (mw, nw):
            if device:
                pri = prev_b.priors_device()
                assert all(p.on_device() for p in pri)
                blocks = [tcv.shifted_prior_blocks(pri[k], prev_W[k]) for k in range(B)]
                del prev_b                                     # the handles keep the batch's result buffer alive
                Wk = [tcv.Window(dict(frame[k], prior=dict(blocks=blocks[k])), prior=pri[k]) for k in range(B)]
            else:
                prev_b.download_priors(compact=True)
                pri = prev_b.priors()
                ds = []
                for k in range(B):
                    d = pri[k].export(); d["blocks"] = tcv.shifted_prior_blocks(pri[k], prev_W[k]); ds.append(d)
                del prev_b
                Wk = [tcv.Window(dict(frame[k], prior=ds[k])) for k in range(B)]
            bk = _marg_batch(tcv, frame, Wk)
            s = _run(bk, tcv)
            assert list(bk.marg_status()) == [0] * B
            out.append((s, [w.states() for w in Wk]))
            prev_b, prev_W = bk, Wk
        last = prev_b.priors_device() if device else (prev_b.download_priors(compact=True), prev_b.priors())[1]
        fin = [p.export() for p in last]                       # (materialises the device-resident ones)
        assert not any(p.on_device() for p in last)
        return out, fin

    host, fin_h = chain(False)
    dev, fin_d = chain(True)
    for (sh, xh), (sd, xd) in zip(host, dev):
        for k in range(B):
            _same_summary(sh[k], sd[k])
            for key in ("pose", "sb", "ex", "lam"):
                assert np.array_equal(xh[k][key], xd[k][key]), (k, key)
    for k in range(B):
        for key in ("J0", "r0"):
            assert np.array_equal(fin_h[k][key], fin_d[k][key]), (k, key)
        assert fin_h[k]["idx"] == fin_d[k]["idx"] and fin_h[k]["sizes"] == fin_d[k]["sizes"] and (fin_h[k]["m"], fin_h[k]["n"]) == (fin_d[k]["m"], fin_d[k]["n"])
        assert all(np.array_equal(a, c) for a, c in zip(fin_h[k]["x0"], fin_d[k]["x0"]))


def test_device_and_host_priors_mix_in_one_batch_and_rebind(gpu):
    """a batch whose windows hold device-resident priors, host priors and no prior at all; tcv_problem_set_marginalization_prior hands an
    existing problem the next prior (same layout) without rebuilding it -- same bits as fresh problems"""
    tcv = gpu
    B = 6
    pre = synth.make_windows(8900, B, frame_shift=-1)
    pw = [synth.window_at(pre, k) for k in range(B)]
    mw = [synth.window_at(synth.make_windows(8900, B), k) for k in range(B)]
    W0 = [tcv.Window(w) for w in pw]
    b0 = _marg_batch(tcv, pw, W0); _run(b0, tcv)
    pd = b0.priors_device()
    b0.download_priors(compact=True); ph = b0.priors()
    blocks = [tcv.shifted_prior_blocks(pd[k], W0[k]) for k in range(B)]
    ref = [tcv.Window(dict(mw[k], prior=dict(ph[k].export(), blocks=blocks[k]))) for k in range(B)]
    sr = _run(_marg_batch(tcv, mw, ref), tcv)
    mixed = []
    for k in range(B):
        if k % 3 == 0:
            mixed.append(tcv.Window(dict(mw[k], prior=dict(blocks=blocks[k])), prior=pd[k]))           # device-resident
        elif k % 3 == 1:
            mixed.append(tcv.Window(dict(mw[k], prior=dict(ph[k].export(), blocks=blocks[k]))))       # host
        else:
            mixed.append(tcv.Window(dict(mw[k], prior=None)))                                         # none
    bm = _marg_batch(tcv, mw, mixed); sm = _run(bm, tcv)
    for k in range(B):
        if k % 3 != 2:
            _same_summary(sr[k], sm[k])
            assert np.array_equal(ref[k].pose, mixed[k].pose) and np.array_equal(ref[k].lam, mixed[k].lam)
    # rebind: the problems of `ref` (host priors) take the device-resident priors; a prior with another layout is refused
    for k in range(B):
        ref[k].pose[:] = mw[k]["pose"]; ref[k].sb[:] = mw[k]["speedbias"]; ref[k].ex[:] = mw[k]["ex_pose"]; ref[k].lam[:] = mw[k]["lam"]
        ref[k].set_prior(pd[k])
    s2 = _run(_marg_batch(tcv, mw, ref), tcv)
    for k in range(B):
        _same_summary(sr[k], s2[k])
    other = bm.priors_device()[2]      # the prior of a window that had none: n = 75 too, but made from other blocks' values -- same layout is accepted
    ref[0].set_prior(other)
    with pytest.raises(tcv.TcvError):
        ref[0].set_prior(tcv.Prior.from_dict(dict(m=0, n=6, sizes=[7], idx=[0], x0=[np.zeros(7)], J0=np.eye(6), r0=np.zeros(6))))


def test_native_estimator_with_device_resident_priors_matches_the_host_round_trip(gpu):
    """include/tcv_estimator.h keeps last_marginalization_info on the device (default); TCV_EST_HOST_PRIORS=1 is round 3's host round trip:
    the same trajectories bit for bit, both marginalisation modes, association in the loop"""
    streams = [replay.simulate_stream(44, 30, max_features=30, associate=True),
               replay.simulate_stream_euroc("V2_02_medium", 30, start_s=1.0, max_features=40, max_lines=5, associate=True)]
    try:
        dev = replay.run_many_native(streams, num_iterations=8)
        os.environ["TCV_EST_HOST_PRIORS"] = "1"
        host = replay.run_many_native(streams, num_iterations=8)
    finally:
        os.environ.pop("TCV_EST_HOST_PRIORS", None)
    for a, c in zip(dev, host):
        assert len(a["t"]) == len(c["t"]) == 30 - replay.WINDOW_SIZE
        assert {l["flag"] for l in a["log"]} == {0, 1}
        assert np.array_equal(a["p"], c["p"]) and np.array_equal(a["q"], c["q"]) and np.array_equal(a["v"], c["v"])
        assert [l["iterations"] for l in a["log"]] == [l["iterations"] for l in c["log"]]


def test_device_resident_preintegrations_are_bit_identical_to_the_host_round_trip(gpu):
    """tcv_preintegrate_device: `pre_integrations[]` stay in HBM (estimator.cpp:200-206 keeps them alive between frames).  A handle exports the
    bits of tcv_preintegrate; a window whose IMU factors are ALL handles uploads none of their constants (n_imu x 287 doubles spliced
    device-to-device), its marginalisation problem reads its factor from the solve batch's pool; a window that mixes handles and host
    constants materialises the handles -- every variant gives the bits of the all-host window, solve and marginalisation."""
    tcv = gpu
    B = 4
    batch = synth.make_windows(9100, B)
    wins = [synth.window_at(batch, k) for k in range(B)]
    acc = batch["imu"]["acc"].reshape(-1, synth.IMU_RATE_SUB + 1, 3); gyr = batch["imu"]["gyr"].reshape(-1, synth.IMU_RATE_SUB + 1, 3)
    n = acc.shape[0]
    nper = n // B
    z = np.zeros((n, 3))
    noise = (synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W)
    host = tcv.preintegrate(acc, gyr, synth.DT_IMU, z, z, noise)
    hd = tcv.preintegrate_device(acc, gyr, synth.DT_IMU, z, z, noise)
    for k in range(0, n, 7):
        e = hd[k].export()
        assert hd[k].sum_dt() == host["sum_dt"][k] == e["sum_dt"]
        for key in ("delta_p", "delta_q", "delta_v", "lin_ba", "lin_bg", "jacobian", "covariance"):
            assert np.array_equal(e[key], host[key][k]), key
    # the windows with the device-computed pre-integrations as HOST constants (the reference case) ...
    keys = ("delta_p", "delta_q", "delta_v", "lin_ba", "lin_bg", "sum_dt", "jacobian", "covariance")
    hw = []
    for k in range(B):
        w = dict(wins[k]); im = dict(w["imu"])
        for key in keys:
            im[key] = host[key][k * nper:(k + 1) * nper]
        w["imu"] = im; hw.append(w)

    def run(imu_dev):
        W = [tcv.Window(hw[k], imu_device=None if imu_dev is None else imu_dev(k)) for k in range(B)]
        MW = [tcv.margin_old_window(w) for w in hw]
        M = [tcv.Window(MW[k], share=W[k], imu_device=None if imu_dev is None else [imu_dev(k)[0]]) for k in range(B)]      # MARGIN_OLD: the factor (0, 1)
        b = tcv.Batch(W, M, [tcv.margin_old_drops(W[k], MW[k]) for k in range(B)])
        s = _run(b, tcv)
        b.download_priors(compact=True)
        return s, [w.states() for w in W], [p.export() for p in b.priors()], b.stats()["input_bytes"]

    s0, x0, p0, bytes0 = run(None)
    s1, x1, p1, bytes1 = run(lambda k: hd[k * nper:(k + 1) * nper])                                             # all device-resident
    s2, x2, p2, bytes2 = run(lambda k: [h if i % 2 else None for i, h in enumerate(hd[k * nper:(k + 1) * nper])])  # mixed
    assert bytes0 - bytes1 == 8.0 * B * nper * 287 and bytes2 == bytes0      # nothing of the factors' constants is uploaded
    for s, x, p in ((s1, x1, p1), (s2, x2, p2)):
        for k in range(B):
            _same_summary(s0[k], s[k])
            for key in ("pose", "sb", "ex", "lam"):
                assert np.array_equal(x0[k][key], x[k][key]), (k, key)
            assert np.array_equal(p0[k]["J0"], p[k]["J0"]) and np.array_equal(p0[k]["r0"], p[k]["r0"])


def test_native_estimator_with_device_resident_preintegrations_matches_the_host_round_trip(gpu):
    """include/tcv_estimator.h keeps pre_integrations[] on the device (default); TCV_EST_HOST_PREINT=1 brings them to the host and uploads
    them with every window: the same trajectories bit for bit (MARGIN_OLD shifts the handles, MARGIN_SECOND_NEW re-integrates a merged buffer)"""
    streams = [replay.simulate_stream(45, 30, max_features=30),
               replay.simulate_stream_euroc("V1_02_medium", 30, start_s=1.0, max_features=40, max_lines=5, associate=True)]
    try:
        dev = replay.run_many_native(streams, num_iterations=8)
        os.environ["TCV_EST_HOST_PREINT"] = "1"
        host = replay.run_many_native(streams, num_iterations=8)
    finally:
        os.environ.pop("TCV_EST_HOST_PREINT", None)
    for a, c in zip(dev, host):
        assert len(a["t"]) == len(c["t"]) == 30 - replay.WINDOW_SIZE and {l["flag"] for l in a["log"]} == {0, 1}
        assert np.array_equal(a["p"], c["p"]) and np.array_equal(a["q"], c["q"]) and np.array_equal(a["v"], c["v"])


def test_device_memory_returns_to_its_level_after_the_objects_are_gone(gpu):
    """tcv_device_memory_stats: batches, device-resident priors (they keep a batch's result buffer alive) and pre-integration handles give
    every byte back when the last owner goes -- chains of frames through the batch API and through the native estimator leave the allocator's live bytes / buffer count where they were"""
    import gc
    tcv = gpu
    B = 6
    pre = synth.make_windows(9300, B, frame_shift=-1)
    pw = [synth.window_at(pre, k) for k in range(B)]
    mw = [synth.window_at(synth.make_windows(9300, B), k) for k in range(B)]
    streams = [replay.simulate_stream(46, 24, max_features=30)]

    def cycle():
        W = [tcv.Window(w) for w in pw]
        b = _marg_batch(tcv, pw, W); _run(b, tcv)
        for _ in range(4):      # frame-to-frame hand-over on the device; every batch but the last is dropped while its priors are in use
            pri = b.priors_device()
            blocks = [tcv.shifted_prior_blocks(pri[k], W[k]) for k in range(B)]
            W = [tcv.Window(dict(mw[k], prior=dict(blocks=blocks[k])), prior=pri[k]) for k in range(B)]
            b = _marg_batch(tcv, mw, W); _run(b, tcv)
        kept = b.priors_device()[0]
        del b, W, pri
        gc.collect()
        assert kept.on_device() and tcv.device_memory_stats()[0] > 0      # one handle alone keeps its batch's result buffer
        kept.export()
        del kept
        batch = synth.make_windows(9301, 2)
        acc = batch["imu"]["acc"].reshape(-1, synth.IMU_RATE_SUB + 1, 3); gyr = batch["imu"]["gyr"].reshape(-1, synth.IMU_RATE_SUB + 1, 3)
        z = np.zeros((acc.shape[0], 3))
        hd = tcv.preintegrate_device(acc, gyr, synth.DT_IMU, z, z, (synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W))
        del hd
        replay.run_many_native(streams, num_iterations=4)
        gc.collect()

    cycle()
    live0, _, n0 = tcv.device_memory_stats()
    for _ in range(2):
        cycle()
    live1, cached1, n1 = tcv.device_memory_stats()
    assert (live1, n1) == (live0, n0), (live0, n0, live1, n1)
    assert cached1 <= 3 << 30


@pytest.mark.parametrize("B", [4, 300])
def test_priors_handed_on_without_waiting_for_the_marginalisation_give_the_same_bits(gpu, B):
    """tcv_batch_get_priors_device_async: the handles are taken while the marginalisation may still be running; the next tcv_batch_create is
    ordered behind it by the result buffer's event (the batch calls below run on the default stream, the upload and the splice on the
    calling thread's own stream: the cross-stream case), the number of thresholded rows is read on the device -- the chain of frames gives
    the bits of the chain that waits (tcv_batch_get_priors_device); the statuses are asked for afterwards"""
    tcv = gpu
    pre = synth.make_windows(9400, B, frame_shift=-1)
    pw = [synth.window_at(pre, k) for k in range(B)]
    main = synth.make_windows(9400, B)
    mw = [synth.window_at(main, k) for k in range(B)]
    opts = tcv.default_options(8, True)

    def chain(nowait):
        W = [tcv.Window(w) for w in pw]
        b = _marg_batch(tcv, pw, W)
        out, old = [], []
        for frame in range(3):
            b.solve(opts); b.gauge_fix(); b.download_states(); s = b.summaries()      # (waits for the solve: the frame's result)
            b.marginalize()
            pri = b.priors_device(nowait=nowait)
            if nowait:
                assert all(p.on_device() for p in pri)
            out.append((s, [w.states() for w in W]))
            blocks = [tcv.shifted_prior_blocks(pri[k], W[k]) for k in range(B)]
            old.append(b)                                     # (kept until its status has been read)
            W = [tcv.Window(dict(mw[k], prior=dict(blocks=blocks[k])), prior=pri[k]) for k in range(B)]
            b = _marg_batch(tcv, mw, W)
        for ob in old:
            assert list(ob.marg_status()) == [0] * B
        b.solve(opts); b.synchronize(); b.download_states()
        out.append((b.summaries(), [w.states() for w in W]))
        return out, [p.export() for p in pri]

    ref, fin_r = chain(False)
    got, fin_g = chain(True)
    for (sr, xr), (sg, xg) in zip(ref, got):
        for k in range(B):
            _same_summary(sr[k], sg[k])
            for key in ("pose", "sb", "ex", "lam"):
                assert np.array_equal(xr[k][key], xg[k][key]), (k, key)
    for k in range(B):
        assert np.array_equal(fin_r[k]["J0"], fin_g[k]["J0"]) and np.array_equal(fin_r[k]["r0"], fin_g[k]["r0"])
