"""Developer tool: per-phase cycles of the marginalisation kernel on the bench workload (windows WITH a prior).
Run with TCV_LIB=tc-viml_amd/libtcv_hip_prof.so TCV_DEBUG=1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, ROOT)
import synth, tcv, bench
B = 8
batch, wins, keep = bench.build_batches(tcv, synth, 100000, B)
o = tcv.default_options(8, True)
print("---- main windows (with prior) ----", file=sys.stderr, flush=True)
batch.solve(o); batch.marginalize(); batch.synchronize()
print("stats", batch.stats())
P = batch.prior(0)
print(P.dims())
