"""Soak test (GPU): long lock-step replays with several feature densities, with and without the 2D-3D association in the loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd"))
import numpy as np
import replay, ate
F = int(sys.argv[1]) if len(sys.argv) > 1 else 120
for feats, assoc in ((20, False), (40, True), (70, False), (110, True)):
    streams = [replay.simulate_stream(1000 + 7 * k + feats, F, max_features=feats, associate=assoc) for k in range(6)]
    t0 = time.perf_counter()
    try:
        outs = replay.run_many(streams, replay.HipBackend(), num_iterations=8)
    except Exception as e:
        print("features", feats, "associate", assoc, "FAILED:", repr(e)[:300]); continue
    errs = []
    for st, o in zip(streams, outs):
        i, j = ate.associate(o["t"], st["t"]); errs.append(ate.ate_rmse(o["p"][i], st["gt_p"][j]))
    flags = sum(l["flag"] for o in outs for l in o["log"]); tot = sum(len(o["log"]) for o in outs)
    print({"features": feats, "associate": assoc, "frames": tot, "second_new": flags, "max_landmarks": max(l["n_landmarks"] for o in outs for l in o["log"]),
           "aligned_ate_median_m": round(float(np.median(errs)), 3), "aligned_ate_max_m": round(float(np.max(errs)), 3), "s": round(time.perf_counter() - t0, 1)})
