"""Developer tool (GPU): run-to-convergence statistics on the benchmark batch (Ceres tolerances, max 100 iterations)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, ROOT)
import numpy as np
import synth, tcv, bench
B = 1024
batch, wins, keep = bench.build_batches(tcv, synth, 100000, B)
o = tcv.default_options(100, False)
for rep in range(3):
    batch.solve(o); batch.synchronize()
s = batch.summaries()
it = np.array([s[k].num_iterations - 1 for k in range(B)]); term = np.array([s[k].termination for k in range(B)])
print("solve_ms", round(batch.stats()["solve_ms"], 3), "iterations mean/median/max", round(float(it.mean()), 2), int(np.median(it)), int(it.max()),
      "termination counts", {int(t): int((term == t).sum()) for t in np.unique(term)})
o8 = tcv.default_options(8, True); batch.solve(o8); batch.synchronize(); s8 = batch.summaries()
fc = np.array([s[k].final_cost for k in range(B)]); f8 = np.array([s8[k].final_cost for k in range(B)])
print("final cost after 8 fixed iterations vs converged: median ratio", round(float(np.median(f8 / fc)), 6), "max", round(float((f8 / fc).max()), 4))
