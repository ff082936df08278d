"""Developer check: ESTIMATE_TD windows, chain layout against dense layout over many seeds, with and without a prior that holds Td
(the prior comes from the GPU marginalisation of the same window).   python tests/dev/td_sweep.py [n_seeds]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth, tcv as gpu
from util import rel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ws = [synth.with_time_offset(synth.window_at(synth.make_windows(1000 + k, 1), 0), 1000 + k, TR=0.02 if k % 2 else 0.0) for k in range(n)]


def priors_for(ws):
    """MARGIN_OLD of every window at its initial states -> prior dicts (n = 76, Td last) re-based on the next window's blocks"""
    out = []
    mws = [gpu.margin_old_window(w) for w in ws]
    Wm = [gpu.Window(m) for m in mws]
    b = gpu.Batch(Wm, Wm, [gpu.margin_old_drops(W, m) for W, m in zip(Wm, mws)])
    b.marginalize(); b.synchronize()
    for k, W in enumerate(Wm):
        P = b.prior(k); d = P.export(); d["blocks"] = gpu.shifted_prior_blocks(P, W)
        out.append(d)
    return out


def run(ws, variant):
    gpu.check(gpu.lib().tcv_set_solver_variant(variant))
    try:
        Ws = [gpu.Window(w) for w in ws]
        b = gpu.Batch(Ws)
        lay = b.plan_stats()["layout"]
        b.solve(gpu.default_options(8, True)); b.synchronize(); b.download_states()
        s = b.summaries()
        return lay, [([s[k].step_ok[i] for i in range(9)], [s[k].dogleg_case[i] for i in range(9)], s[k].final_cost, Ws[k].pose.copy(), Ws[k].sb.copy(), float(Ws[k].td[0]), 0 if s[k].termination != 5 else 1) for k in range(len(ws))]
    finally:
        gpu.check(gpu.lib().tcv_set_solver_variant(0))


for label, batch in (("no prior", ws), ("prior with Td", None)):
    if batch is None:
        pr = priors_for(ws)
        batch = []
        for k, w in enumerate(ws):
            nxt = ws[(k + 1) % n]
            d = dict(pr[k])
            d["x0"] = [np.array({"pose": nxt["pose"], "sb": nxt["speedbias"]}[nm][i], dtype=float).copy() if nm in ("pose", "sb")
                       else (np.array(nxt["ex_pose"], dtype=float).copy() if nm == "ex" else np.array([float(nxt.get("td", 0.0))])) for nm, i in d["blocks"]]
            batch.append(dict(nxt, prior=d))
    la, A = run(batch, 0)
    ld, D = run(batch, 1)
    worst = np.zeros(4); ntr = 0; bad = 0
    for a, d in zip(A, D):
        ntr += int(a[0] == d[0] and a[1] == d[1]); bad += int(a[6] != 0 or d[6] != 0)
        worst = np.maximum(worst, [abs(a[2] - d[2]) / d[2], rel(a[3], d[3]), rel(a[4], d[4]), abs(a[5] - d[5]) / max(1e-3, abs(d[5]))])
    print(f"{label:14s} {n} windows, layouts {la} / {ld}: identical traces {ntr}, failed solves {bad}, worst rel. diff cost {worst[0]:.1e} pose {worst[1]:.1e} speed-bias {worst[2]:.1e} td {worst[3]:.1e}")
