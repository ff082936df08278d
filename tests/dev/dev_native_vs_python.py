"""native estimator vs replay.Replay (both on the HIP kernels), frame by frame (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tc-viml_amd",):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, replay
assoc = True
streams = [replay.simulate_stream(40 + k, 30, max_features=30, associate=assoc) for k in range(2)]
streams.append(replay.simulate_stream_euroc("V2_02_medium", 30, start_s=1.0, max_features=40, max_lines=5, associate=assoc))
nat = replay.run_many_native(streams, num_iterations=8)
py = replay.run_many(streams, replay.HipBackend(), num_iterations=8)
for a, b in zip(nat, py):
    d = np.linalg.norm(a["p"] - b["p"], axis=1)
    print("dp", np.array2string(d, precision=1))
    for key in ("n_line", "n_line_obs", "flag", "n_proj"):
        print(key, [l[key] for l in a["log"]], [l[key] for l in b["log"]] if [l[key] for l in a["log"]] != [l[key] for l in b["log"]] else "same")
