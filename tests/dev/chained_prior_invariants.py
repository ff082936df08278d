"""Developer check (GPU): what the difference between two correct priors (HIP-made vs oracle-made, A' 5e-7 apart) does to the NEXT solve,
quantity by quantity -- the first trust-region step's well-posed scalars against the end state after eight iterations
(tests/test_gpu_marg.py::test_prior_round_trip_and_chained_solve gates on the former)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ("tc-viml_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np
import orc, synth, tcv
from util import golden_windows, rel, fro

pre, main, z = golden_windows()
W = [tcv.Window(pre)]; mw = tcv.margin_old_window(pre); M = [tcv.Window(mw, share=W[0])]; dr = [tcv.margin_old_drops(W[0], mw)]
b = tcv.Batch(W, M, dr); b.solve(tcv.default_options(8, True)); b.marginalize(); b.synchronize()
P = b.prior(0); d = P.export(); d["blocks"] = tcv.shifted_prior_blocks(P, W[0])
w = dict(main); w["prior"] = d
Wn = tcv.Window(w)
bn = tcv.Batch([Wn]); bn.solve(tcv.default_options(8, True, record_first_step=True)); bn.synchronize(); bn.download_states()
s = bn.summaries()[0]
O = orc.Window(main); so = O.solve(8, True)
r = lambda a, c: abs(a - c) / abs(c)
print("prior pair: J0'J0 %.2e J0'r0 %.2e" % (fro(d["J0"].T @ d["J0"], main["prior"]["J0"].T @ main["prior"]["J0"]), fro(d["J0"].T @ d["r0"], main["prior"]["J0"].T @ main["prior"]["r0"])))
print("initial cost %.3e | model cost change of step 1 %.3e | cost after step 1 %.3e | step norm 1 %.3e | radius after step 1 %.3e"
      % (r(s.initial_cost, so.initial_cost), r(s.model_cost_change[1], so.model_cost_change[1]), r(s.cost[1], so.cost[1]), r(s.step_norm[1], so.step_norm[1]), r(s.radius[1], so.radius[1])))
for i in range(1, 9):
    print("  iteration %d: cost %.3e model %.3e step norm %.3e ok %d/%d case %d/%d" % (i, r(s.cost[i], so.cost[i]), r(s.model_cost_change[i], so.model_cost_change[i]) if so.model_cost_change[i] else -1,
                                                                               r(s.step_norm[i], so.step_norm[i]), s.step_ok[i], so.step_ok[i], s.dogleg_case[i], so.dogleg_case[i]))
print("end state: cost %.3e pose %.3e sb %.3e | max |dp| %.3e m" % (r(s.final_cost, so.final_cost), rel(Wn.pose, O.states()["pose"]), rel(Wn.sb, O.states()["sb"]),
                                                                 np.abs(Wn.pose[:, :3] - O.states()["pose"][:, :3]).max()))
