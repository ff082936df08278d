"""Developer tool (GPU): how many windows of the bench batch take the Jacobi safety net in the marginalisation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, ROOT)
import numpy as np
import synth, tcv, bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for variant in (0, 1):
    tcv.lib().tcv_set_solver_variant(variant)
    batch, wins, keep = bench.build_batches(tcv, synth, 100000, B)
    o = tcv.default_options(8, True)
    for gf in (False, True):
        batch.solve(o)
        if gf: batch.gauge_fix()
        batch.marginalize(); batch.synchronize()
        st = batch.marg_status()
        print("variant", variant, "gauge_fix", gf, "status counts", {int(k): int((st == k).sum()) for k in np.unique(st)}, batch.stats())
        for k in np.nonzero(st == 2)[0][:3]:
            os.environ["TCV_DEBUG"] = "1"
            P = batch.prior(int(k)); d = P.export(); As, bs = P.schur()
            del os.environ["TCV_DEBUG"]
            lam = np.linalg.eigvalsh(As)
            print("window", k, "eig(A') :", np.array2string(lam, precision=4, max_line_width=250))
