"""Full-length EuRoC-trajectory replays (BASELINE configs[3]): every sequence the reference ships ground truth for, the whole 36 s
excerpt, HIP back end vs the CPU oracle back end.  Per sequence and line mode: identical decisions?, largest position difference,
both ATEs against the ground truth.  Writes gpurun_out/euroc_full.json (copied to profiles/r02_euroc_full.json).
`python tests/dev/replay_euroc_full.py` on the GPU box."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tc-viml_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import replay, ate, orc
from replay_oracle import OracleBackend

orc.build()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 355
rows = []
for mode in ("none", "given", "associate"):
    for seq in replay.EUROC_SEQUENCES:
        st = replay.simulate_stream_euroc(seq, F, start_s=0.5, max_features=60, max_lines=0 if mode == "none" else 8, associate=mode == "associate")
        t0 = time.perf_counter(); hip = replay.run(st, replay.HipBackend(), num_iterations=8)
        t1 = time.perf_counter(); ref = replay.run(st, OracleBackend(), num_iterations=8)
        t2 = time.perf_counter()
        d = np.linalg.norm(hip["p"] - ref["p"], axis=1)
        i, j = ate.associate(hip["t"], st["t"])
        same = all([a[k] for a in hip["log"]] == [b[k] for b in ref["log"]] for k in ("flag", "n_line", "n_proj", "iterations"))
        first_mm = int(np.argmax(d > 1e-3)) if (d > 1e-3).any() else -1
        rows.append(dict(mode=mode, seq=seq, frames=len(hip["t"]), seconds=round(float(hip["t"][-1] - hip["t"][0]), 1), same_decisions=bool(same),
                         max_dp_m=float(d.max()), first_frame_beyond_1mm=first_mm, ate_hip_m=ate.ate_rmse(hip["p"][i], st["gt_p"][j]),
                         ate_oracle_m=ate.ate_rmse(ref["p"][i], st["gt_p"][j]), hip_s=round(t1 - t0, 1), oracle_s=round(t2 - t1, 1)))
        print(json.dumps(rows[-1]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "euroc_full.json"), "w") as f:
    json.dump(rows, f, indent=1)
