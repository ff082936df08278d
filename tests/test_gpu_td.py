"""T1 in the solver: ESTIMATE_TD windows -- every point factor a ProjectionTdFactor on the extra 1-dim block
para_Td[0] (estimator.cpp:1703-1707, :1757-1763) -- solved (chain layout, the default, and dense layout of the fused solver) and
marginalised by the HIP path against the NumPy restatement (oracle/np_oracle.py)."""
import numpy as np
import pytest

import np_oracle as NO
import synth
from util import fro, rel

pytestmark = pytest.mark.gpu

CASE = {"GAUSS_NEWTON": 1, "CAUCHY": 2, "DOGLEG": 3}


def td_window(seed, **kw):
    return synth.with_time_offset(synth.window_at(synth.make_windows(seed, 1), 0), seed, **kw)


@pytest.fixture(params=["chain", "dense"])
def layout(request, gpu):
    """chain: Td is the last column of the pose part, the speed-bias chain is eliminated in front of it (the default);
    dense: solver variant 1, one 172-dim system"""
    gpu.check(gpu.lib().tcv_set_solver_variant(0 if request.param == "chain" else 1))
    yield request.param
    gpu.check(gpu.lib().tcv_set_solver_variant(0))


@pytest.mark.parametrize("seed,TR", [(21, 0.0), (22, 0.02)])
def test_td_window_solve_vs_oracle(gpu, layout, seed, TR):
    w = td_window(seed, TR=TR)
    W = gpu.Window(w)
    b = gpu.Batch([W])
    assert b.plan_stats()["layout"] == layout
    b.solve(gpu.default_options(8, True)); b.synchronize(); b.download_states()
    s = b.summaries()[0]
    P = NO.Problem(w)
    x, so = NO.solve(P, 8, True)
    its = so["iterations"]
    assert s.num_iterations == len(its)
    assert abs(s.initial_cost - so["initial_cost"]) < 1e-9 * so["initial_cost"]
    assert abs(s.final_cost - so["final_cost"]) < 1e-6 * so["final_cost"]
    assert [s.step_ok[i] for i in range(1, len(its))] == [int(it["step_ok"]) for it in its[1:]]
    assert rel(W.pose, x["pose"]) < 1e-6 and rel(W.sb, x["sb"]) < 1e-6 and rel(W.lam, x["lam"]) < 1e-5
    assert abs(W.td[0] - x["td"][0]) < 1e-6 * max(1e-3, abs(x["td"][0]))
    assert 0.001 < W.td[0] < 0.006                                   # moving from 0 towards the true offset of 4 ms
    # plain ProjectionFactors on the same data give another answer: the Td column is really in the system
    w0 = dict(w); w0.pop("td")
    W0 = gpu.Window(w0); b0 = gpu.Batch([W0]); b0.solve(gpu.default_options(8, True)); b0.synchronize()
    assert b0.summaries()[0].final_cost > 1.05 * s.final_cost


def test_td_window_marginalisation_vs_oracle(gpu, layout):
    """MARGIN_OLD with ProjectionTdFactors among the marginalised factors (estimator.cpp:1962-1972): para_Td is a kept block, so
    the new prior has n = 76 and a size-1 block at the end; then the chained window uses that prior (kind 3 = Td)."""
    w = td_window(23)
    P = NO.Problem(w)
    x, _ = NO.solve(P, 8, True)
    w2 = dict(w, pose=x["pose"], speedbias=x["sb"], ex_pose=x["ex"], lam=x["lam"], td=float(x["td"][0]))
    po, dbg = NO.marginalize_old(NO.Problem(w2), NO.Problem(w2).x0())
    mw = gpu.margin_old_window(w2)
    Wm = gpu.Window(mw)
    b = gpu.Batch([Wm], [Wm], [gpu.margin_old_drops(Wm, mw)])
    b.marginalize(); b.synchronize()
    Pr = b.prior(0); d = Pr.export(); As, bs = Pr.schur()
    assert (d["m"], d["n"]) == (po["m"], po["n"]) and d["n"] == 76 and d["sizes"] == po["sizes"] and d["sizes"][-1] == 1
    assert gpu.shifted_prior_blocks(Pr, Wm) == [tuple(bk) for bk in po["blocks"]]
    assert fro(As, dbg["A_schur"]) < 1e-5 and fro(bs, dbg["b_schur"]) < 1e-6
    assert fro(d["J0"].T @ d["J0"], dbg["A_schur"]) < 1e-5
    # the next window with this prior attached: both sides solve it and agree
    nxt = synth.with_time_offset(synth.window_at(synth.make_windows(24, 1), 0), 24)
    keep = dict(po, blocks=[tuple(bk) for bk in po["blocks"]])
    x0 = []
    for (nm, i), v in zip(keep["blocks"], po["x0"]):
        cur = {"pose": nxt["pose"], "sb": nxt["speedbias"]}.get(nm)
        x0.append(np.array(cur[i], dtype=float).copy() if cur is not None else (np.array(nxt["ex_pose"], dtype=float).copy() if nm == "ex" else np.array([0.0])))
    keep["x0"] = x0                                                   # linearised at the new window's own initial states
    nxt = dict(nxt, prior=keep)
    W = gpu.Window(nxt); b2 = gpu.Batch([W]); b2.solve(gpu.default_options(8, True)); b2.synchronize(); b2.download_states()
    assert b2.plan_stats()["layout"] == layout                        # (chain: the prior ties Td to the first speed-bias block)
    s = b2.summaries()[0]
    x2, so = NO.solve(NO.Problem(nxt), 8, True)
    assert abs(s.final_cost - so["final_cost"]) < 1e-6 * so["final_cost"]
    assert rel(W.pose, x2["pose"]) < 1e-6 and abs(W.td[0] - x2["td"][0]) < 1e-6 * max(1e-3, abs(x2["td"][0]))


def test_td_batch_chain_and_dense_agree(gpu):
    """a batch of ESTIMATE_TD windows larger than the chip (two chain-layout workgroups per CU): the chain kernel with ProjectionTdFactor
    and the dense kernel solve the same systems -- identical accept / reject and dogleg sequences, states within 1e-8"""
    ws = [td_window(100 + k, TR=0.02 if k % 2 else 0.0) for k in range(8)]
    out = {}
    for variant in (0, 1):
        gpu.check(gpu.lib().tcv_set_solver_variant(variant))
        try:
            Ws = [gpu.Window(ws[k % 8]) for k in range(300)]
            b = gpu.Batch(Ws)
            assert b.plan_stats()["layout"] == ("chain", "dense")[variant]
            b.solve(gpu.default_options(8, True)); b.synchronize(); b.download_states()
            s = b.summaries()
            out[variant] = [([s[k].step_ok[i] for i in range(9)], [s[k].dogleg_case[i] for i in range(9)], s[k].final_cost,
                             Ws[k].pose.copy(), Ws[k].sb.copy(), float(Ws[k].td[0])) for k in range(300)]
        finally:
            gpu.check(gpu.lib().tcv_set_solver_variant(0))
    worst = np.zeros(4)
    for a, d in zip(out[0], out[1]):
        assert a[0] == d[0] and a[1] == d[1]
        worst = np.maximum(worst, [abs(a[2] - d[2]) / d[2], rel(a[3], d[3]), rel(a[4], d[4]), abs(a[5] - d[5]) / max(1e-3, abs(d[5]))])
    print("chain vs dense, ESTIMATE_TD windows: cost %.1e pose %.1e speed-bias %.1e td %.1e" % tuple(worst))
    assert worst[0] < 1e-7 and worst[1] < 1e-7 and worst[2] < 1e-6 and worst[3] < 1e-6      # measured 2e-9 on the cost
    for k in range(8, 300):                                           # the same window gives the same bits wherever it runs
        assert out[0][k][2] == out[0][k % 8][2] and np.array_equal(out[0][k][3], out[0][k % 8][3])


def test_mixed_batch_td_and_plain_windows(gpu):
    """one chain-layout batch holding ESTIMATE_TD windows and plain windows (two plans, the kernel instance with ProjectionTdFactor runs
    both): every window ends where it ends in a batch of its own (to rounding: a plain window alone takes the cooperative plan, whose
    chunks -- the order of the sums -- differ), and twice the same window in the batch gives twice the same bits"""
    wt = td_window(31, TR=0.02)
    wp = dict(td_window(32)); wp.pop("td")
    def solve(ws):
        Ws = [gpu.Window(w) for w in ws]
        b = gpu.Batch(Ws)
        assert b.plan_stats()["layout"] == "chain"
        b.solve(gpu.default_options(8, True)); b.synchronize(); b.download_states()
        return [(s.final_cost, W.pose.copy(), W.sb.copy()) for s, W in zip(b.summaries(), Ws)], b.plan_stats()["num_plans"]
    mixed, nplans = solve([wt, wp, wt, wp])
    assert nplans == 2
    (alone_t,), _ = solve([wt])
    (alone_p,), _ = solve([wp])
    for got, want in ((mixed[0], alone_t), (mixed[2], alone_t), (mixed[1], alone_p), (mixed[3], alone_p)):
        assert abs(got[0] - want[0]) < 1e-11 * want[0] and rel(got[1], want[1]) < 1e-10 and rel(got[2], want[2]) < 1e-10
    for a, b in ((0, 2), (1, 3)):
        assert mixed[a][0] == mixed[b][0] and np.array_equal(mixed[a][1], mixed[b][1]) and np.array_equal(mixed[a][2], mixed[b][2])


def test_td_window_in_cooperative_mode_gives_the_same_bits(gpu):
    """a single ESTIMATE_TD window runs on 1 + H workgroups like every other small batch (the helpers evaluate the ProjectionTdFactors);
    the same plan on one workgroup (the chain kernel instance with ProjectionTdFactor) gives the same bits"""
    w = td_window(41, TR=0.02)
    res = {}
    for wg in (0, 1):
        W = gpu.Window(w); b = gpu.Batch([W])
        assert b.plan_stats()["layout"] == "chain"
        b.solve(gpu.default_options(8, True, workgroups_per_window=wg)); b.synchronize(); b.download_states()
        s = b.summaries()[0]
        res[wg] = (b.cooperative(), s.final_cost, [s.cost[i] for i in range(9)], W.pose.copy(), W.sb.copy(), W.lam.copy(), float(W.td[0]))
    co = res[0][0]
    assert co["helpers"] >= 2 and co["last_solve_workgroups"] == 1 + co["helpers"] and res[1][0]["last_solve_workgroups"] == 1, co
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2] and res[0][6] == res[1][6]
    for k in (3, 4, 5):
        assert np.array_equal(res[0][k], res[1][k])
