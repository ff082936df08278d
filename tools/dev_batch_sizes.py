"""Developer tool: solve / marginalisation kernel times of the benchmark workload at several batch sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, ROOT)
import numpy as np
import synth, tcv, bench
for B in [int(a) for a in sys.argv[1:]] or [1, 16, 256, 1024]:
    batch, wins, keep = bench.build_batches(tcv, synth, 100000, B)
    opts = tcv.default_options(8, True)
    ts, tm = [], []
    for _ in range(8):
        batch.solve(opts); batch.gauge_fix(); batch.marginalize(); batch.synchronize()
        st = batch.stats(); ts.append(st["solve_ms"]); tm.append(st["marg_ms"])
    print("B = %5d: solve %.3f ms, marginalise %.3f ms -> %.0f solves/s (kernels only), layout %s" % (B, np.median(ts[2:]), np.median(tm[2:]), B / (np.median(ts[2:]) + np.median(tm[2:])) * 1e3, batch.plan_stats()), flush=True)
    del batch
