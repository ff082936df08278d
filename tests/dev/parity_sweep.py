"""Developer checker (GPU + oracle): parity sweep of the fused solve + gauge fix + marginalisation over many benchmark windows.
For every window: identical dogleg / accept traces, relative error of the final cost, the states, the first step, A' and b'."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tc-viml_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
sys.path.insert(0, ROOT)
import numpy as np
import np_oracle as NO, orc, synth, tcv, bench
from util import rel, fro

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
batch, wins, keep = bench.build_batches(tcv, synth, 200000, B)
Wm = keep[0]
opts = tcv.default_options(8, True, True, 256, True)
batch.solve(opts); batch.gauge_fix(); batch.marginalize(); batch.synchronize(); batch.download_states()
s = batch.summaries()
st_counts = {int(k): int((batch.marg_status() == k).sum()) for k in np.unique(batch.marg_status())}
worst = dict(cost=0, pose=0, sb=0, lam=0, first=0, A=0, b=0); same_trace = 0
for k in range(B):
    O = orc.Window(wins[k]); so = O.solve(8, True); st = O.states()
    same_trace += int([s[k].dogleg_case[i] for i in range(9)] == [so.dogleg_case[i] for i in range(9)] and [s[k].step_ok[i] for i in range(9)] == [so.step_ok[i] for i in range(9)])
    R0 = NO.q2R(np.asarray(wins[k]["pose"])[0, 3:]); P0 = np.asarray(wins[k]["pose"])[0, :3]
    Rs, Ps, Vs, po = orc.gauge_fix(R0, P0, st["pose"], st["sb"])
    sb = st["sb"].copy(); sb[:, :3] = Vs
    worst["cost"] = max(worst["cost"], abs(s[k].final_cost - so.final_cost) / so.final_cost)
    worst["pose"] = max(worst["pose"], rel(Wm[k].pose, po)); worst["sb"] = max(worst["sb"], rel(Wm[k].sb, sb)); worst["lam"] = max(worst["lam"], rel(Wm[k].lam, st["lam"]))
    w2 = dict(wins[k], pose=po, speedbias=sb, ex_pose=st["ex"], lam=st["lam"])
    pref, dbg = orc.Window(w2).marginalize_old()
    As, bs = batch.prior(k).schur()
    worst["A"] = max(worst["A"], fro(As, dbg["A_schur"])); worst["b"] = max(worst["b"], fro(bs, dbg["b_schur"]))
print({"windows": B, "identical_traces": same_trace, "marg_status": st_counts, **{k: float("%.3g" % v) for k, v in worst.items()}})
