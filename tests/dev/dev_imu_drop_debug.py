import sys
sys.path[:0]=['tc-viml_amd','tests','oracle']
import numpy as np
import orc, synth, tcv
from util import golden_windows, rel
pre, main, z = golden_windows()
for base, name in ((pre, "pre"), (main, "main")):
    im = dict(base["imu"]); sd = np.array(im["sum_dt"], dtype=float).copy(); sd[4] = 11.0; im["sum_dt"] = sd
    w_hip = dict(base, imu=im)
    keep = np.arange(len(sd)) != 4
    w_orc = dict(base, imu={k: (np.asarray(v)[keep] if isinstance(v, np.ndarray) and np.asarray(v).shape[:1] == keep.shape else v) for k, v in base["imu"].items()})
    print(name, {k: np.asarray(v).shape for k, v in w_orc["imu"].items()})
    O = orc.Window(w_orc); so = O.solve(8, True)
    for variant in (0, 1):
        tcv.lib().tcv_set_solver_variant(variant)
        for src, nm in ((w_hip, "dropped by the library"), (w_orc, "dropped by the caller")):
            W = tcv.Window(src); b = tcv.Batch([W]); b.solve(tcv.default_options(8, True)); b.synchronize(); b.download_states(); s = b.summaries()[0]
            print(name, "variant", variant, nm, "layout", b.plan_stats()["layout"])
            for i in range(so.num_iterations):
                print("   it %d cost %.10e / %.10e  ok %d/%d case %d/%d step %.3e/%.3e" % (i, s.cost[i], so.cost[i], s.step_ok[i], so.step_ok[i], s.dogleg_case[i], so.dogleg_case[i], s.step_norm[i], so.step_norm[i]))
    tcv.lib().tcv_set_solver_variant(0)
