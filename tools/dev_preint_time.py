"""Developer tool (GPU): wall time of tcv_preintegrate for n buffers of 20 samples (what a lock-step frame pays once)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd"))
import numpy as np
import replay
be = replay.HipBackend()
rng = np.random.default_rng(1)
for n in (1, 5, 8, 64):
    bufs = [dict(acc=rng.normal(size=(20, 3)) + [0, 0, 9.8], gyr=rng.normal(size=(20, 3)) * 0.1, acc0=np.array([0, 0, 9.8]), gyr0=np.zeros(3), ba=np.zeros(3), bg=np.zeros(3)) for _ in range(n)]
    be.preintegrate(bufs)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); be.preintegrate(bufs); ts.append(time.perf_counter() - t0)
    print("n = %3d buffers x 20 samples: tcv_preintegrate (incl. the Python marshalling) median %.3f ms, min %.3f ms" % (n, 1e3 * np.median(ts), 1e3 * min(ts)))
