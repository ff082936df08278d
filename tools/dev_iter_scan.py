"""Developer tool (GPU): solve-kernel time vs number of trust-region iterations (fixed cost vs per-iteration cost)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, ROOT)
import numpy as np
import synth, tcv, bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tcv.lib().tcv_set_solver_variant(variant)
batch, wins, keep = bench.build_batches(tcv, synth, 100000, B)
for it in (0, 1, 2, 4, 8, 16):
    o = tcv.default_options(it, True)
    ts = []
    for rep in range(4):
        batch.solve(o); batch.synchronize(); ts.append(batch.stats()["solve_ms"])
    print("variant", variant, "iterations", it, "solve_ms", round(min(ts), 3))
