"""Multi-sequence replay throughput (BASELINE configs[4] in spirit): S simulated sequences replayed in lock step through the HIP
back end, one device batch per frame.  Reports frames/s and where the wall time goes (the per-frame window management and the
packing of S problems run on the host in Python/C++; the kernels take a few ms per frame)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd"))
import numpy as np
import replay, ate

NATIVE = "--native" in sys.argv      # window management of include/tcv_estimator.h instead of replay.Replay
sys.argv = [a for a in sys.argv if a != "--native"]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
F = int(sys.argv[2]) if len(sys.argv) > 2 else 30
t0 = time.perf_counter()
streams = [replay.simulate_stream(100 + k, F, max_features=30) for k in range(S)]
t1 = time.perf_counter()
be = replay.HipBackend()
outs = replay.run_many_native(streams, num_iterations=8) if NATIVE else replay.run_many(streams, be, num_iterations=8)
t2 = time.perf_counter()
frames = sum(len(o["t"]) for o in outs)
errs = []
for st, o in zip(streams, outs):
    i, j = ate.associate(o["t"], st["t"])
    errs.append(ate.ate_rmse(o["p"][i], st["gt_p"][j]))
print({"window_management": "native" if NATIVE else "python", "sequences": S, "frames_each": F, "optimised_frames": frames, "simulate_s": round(t1 - t0, 2), "replay_s": round(t2 - t1, 2),
       "frames_per_s": round(frames / (t2 - t1), 1), "aligned_ate_m_median": round(float(np.median(errs)), 4), "aligned_ate_m_max": round(float(np.max(errs)), 4)})
