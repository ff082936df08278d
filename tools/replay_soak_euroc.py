"""Soak test (GPU): many EuRoC-trajectory replays in lock step through the native estimator (include/tcv_estimator.h): 5 sequences x
S seeds, dense front end, association in the loop.  Reports failures (exceptions, NaNs, failure detection) and the ATE spread."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd"))
import numpy as np
import replay, ate
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
F = int(sys.argv[2]) if len(sys.argv) > 2 else 300
feats = int(sys.argv[3]) if len(sys.argv) > 3 else 100
streams = [replay.simulate_stream_euroc(seq, F, start_s=0.5 + 0.35 * k, seed=k, max_features=feats, max_lines=20, associate=(k % 2 == 0))
           for seq in replay.EUROC_SEQUENCES for k in range(S)]
t0 = time.perf_counter()
outs = replay.run_many_native(streams, num_iterations=8)
dt = time.perf_counter() - t0
errs = [ate.ate_rmse(o["p"], st["gt_p"][replay.WINDOW_SIZE:][:len(o["p"])]) for o, st in zip(outs, streams)]
worst = int(np.argmax(errs))
print("worst stream", worst, streams[worst]["seq"], "associate", "map_lines" in streams[worst], "ATE %.3f" % errs[worst], "sorted ATEs", np.round(np.sort(errs)[-6:], 3))
bad = sum(1 for o in outs if not np.all(np.isfinite(o["p"])))
print({"streams": len(streams), "frames": sum(len(o["t"]) for o in outs), "seconds": round(dt, 1), "frames_per_s": round(sum(len(o["t"]) for o in outs) / dt, 1),
       "non_finite": bad, "ate_median_m": round(float(np.median(errs)), 4), "ate_max_m": round(float(np.max(errs)), 4),
       "max_landmarks": max(l["n_landmarks"] for o in outs for l in o["log"]), "max_proj": max(l["n_proj"] for o in outs for l in o["log"])})
