"""Developer checker (GPU): run-to-run determinism of the replay paths and, with TCV_LIB pointing at another build, a bit-level
comparison of the priors two builds produce for the same windows."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tc-viml_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
sys.path.insert(0, ROOT)
import numpy as np
import replay, tcv, synth, bench

def h(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]

streams = [replay.simulate_stream_euroc("V2_02_medium", 30, start_s=1.0, max_features=40, max_lines=5, associate=False)]
for rep in range(2):
    nat = replay.run_many_native(streams, num_iterations=8)
    py = replay.run_many(streams, replay.HipBackend(), num_iterations=8)
    print("run", rep, "native", h(nat[0]["p"]), "python", h(py[0]["p"]), "max |dp| %.2e" % np.linalg.norm(nat[0]["p"] - py[0]["p"], axis=1).max())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 300
batch, wins, keep = bench.build_batches(tcv, synth, 200000, B)
opts = tcv.default_options(8, True, True, 256, True)
for rep in range(2):
    batch.solve(opts); batch.gauge_fix(); batch.marginalize(); batch.synchronize()
    pr = [batch.prior(k) for k in range(0, B, 37)]
    print("batch", B, "rep", rep, "J0", h(np.concatenate([p.export()["J0"].ravel() for p in pr])), "A'", h(np.concatenate([p.schur()[0].ravel() for p in pr])))
