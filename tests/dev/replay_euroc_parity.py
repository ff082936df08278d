"""Full-length EuRoC-trajectory replay, HIP back end vs the CPU oracle back end (BASELINE configs[3]'s "ATE parity vs the
reference"): per sequence the largest position difference between the two replays and both ATEs against the ground truth.
Uses oracle/ => lives under tests/.  `python tests/dev/replay_euroc_parity.py [frames] [associate]` on the GPU box."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tc-viml_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import replay, ate, orc
from replay_oracle import OracleBackend

F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
assoc = len(sys.argv) > 2 and sys.argv[2] == "associate"
orc.build()
rows = []
for seq in replay.EUROC_SEQUENCES:
    st = replay.simulate_stream_euroc(seq, F, start_s=0.5, max_features=60, max_lines=8, associate=assoc)
    t0 = time.perf_counter(); hip = replay.run(st, replay.HipBackend(), num_iterations=8)
    t1 = time.perf_counter(); ref = replay.run(st, OracleBackend(), num_iterations=8)
    t2 = time.perf_counter()
    d = np.linalg.norm(hip["p"] - ref["p"], axis=1)
    i, j = ate.associate(hip["t"], st["t"])
    same = [a["n_line"] for a in hip["log"]] == [b["n_line"] for b in ref["log"]] and [a["flag"] for a in hip["log"]] == [b["flag"] for b in ref["log"]]
    rows.append(dict(seq=seq, frames=len(hip["t"]), max_dp_m=float(d.max()), rms_dp_m=float(np.sqrt((d ** 2).mean())), same_decisions=bool(same),
                     ate_hip_m=ate.ate_rmse(hip["p"][i], st["gt_p"][j]), ate_oracle_m=ate.ate_rmse(ref["p"][i], st["gt_p"][j]),
                     hip_s=round(t1 - t0, 1), oracle_s=round(t2 - t1, 1)))
    print(json.dumps(rows[-1]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "euroc_parity%s.json" % ("_assoc" if assoc else "")), "w") as f:
    json.dump(rows, f, indent=1)
