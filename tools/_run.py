import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, ROOT)
import numpy as np
import synth, tcv, bench
B = 1024
batch, wins, keep = bench.build_batches(tcv, synth, 100000, B)
opts = tcv.default_options(8, True)
def run(mask, reps=6):
    os.environ["TCV_ABLATE_SKIP"] = str(mask | (3 << 30))
    t = []
    for _ in range(reps):
        batch.solve(opts); batch.synchronize(); t.append(batch.stats()["solve_ms"])
    return float(np.median(t[1:]))
ev = (1 << 19) - 1
print("all phases %.3f" % run(0))
print("skeleton (everything skipped) %.3f" % run(ev))
print("skeleton - setup %.3f" % run(ev | (1 << 19)))
print("skeleton - program copies %.3f" % run(ev | (1 << 24)))
print("skeleton - zeroing %.3f" % run(ev | (1 << 25)))
print("skeleton - setup - copies - zeroing %.3f" % run(ev | (1 << 19) | (1 << 24) | (1 << 25)))
print("skeleton - all of these and the chain pipelines/fetch %.3f" % run(ev | (1 << 19) | (1 << 24) | (1 << 25) | (15 << 20)))
