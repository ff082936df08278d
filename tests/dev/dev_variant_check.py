"""Developer check (GPU): chain layout vs dense layout vs the C oracle on the golden window and a few synthetic ones."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tc-viml_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import orc, synth, tcv
from util import golden_windows, rel

pre, main, z = golden_windows()
wins = [pre, main] + [synth.window_at(synth.make_windows(77, 3), k) for k in range(3)]
res = {}
for variant in (0, 1):
    tcv.lib().tcv_set_solver_variant(variant)
    W = [tcv.Window(w) for w in wins]
    b = tcv.Batch(W)
    print("variant", variant, b.plan_stats())
    b.solve(tcv.default_options(8, True, True, 256, True)); b.synchronize(); b.download_states()
    s = b.summaries()
    res[variant] = ([(s[k].final_cost, [s[k].dogleg_case[i] for i in range(9)], [s[k].cost[i] for i in range(9)]) for k in range(len(wins))],
                    [w.pose.copy() for w in W], [w.sb.copy() for w in W], [b.first_step(k) for k in range(len(wins))], b.stats()["solve_ms"])
for k, w in enumerate(wins):
    O = orc.Window(w); so = O.solve(8, True); st = O.states()
    for v in (0, 1):
        fc, dc, costs = res[v][0][k]
        print(k, "variant", v, "final cost rel", abs(fc - so.final_cost) / so.final_cost, "pose", rel(res[v][1][k], st["pose"]), "sb", rel(res[v][2][k], st["sb"]),
              "cases", dc == [so.dogleg_case[i] for i in range(9)], "cost trace", max(abs(costs[i] - so.cost[i]) / so.cost[i] for i in range(9)))
    print(k, "first step chain vs dense", rel(res[0][3][k], res[1][3][k]))
print("solve ms chain/dense (5 windows):", res[0][4], res[1][4])
