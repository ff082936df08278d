"""Developer checker (GPU + oracle): frame-by-frame position difference HIP vs oracle back end on the association replay of
tests/test_gpu_replay.py::test_replay_with_line_association_hip_vs_oracle; TCV_MARG_EIG_MM=1 forces the eigen path for Amm."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tc-viml_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import replay
from replay_oracle import OracleBackend
stream = replay.simulate_stream(int(sys.argv[1]) if len(sys.argv) > 1 else 1, 30, max_features=30, associate=True)
hip = replay.run(stream, replay.HipBackend(), num_iterations=8)
ref = replay.run(stream, OracleBackend(), num_iterations=8)
d = np.linalg.norm(hip["p"] - ref["p"], axis=1)
for k in range(len(d)):
    print(k, "flag", hip["log"][k]["flag"], ref["log"][k]["flag"], "n_line", hip["log"][k]["n_line"], ref["log"][k]["n_line"], "it", hip["log"][k]["iterations"], ref["log"][k]["iterations"],
          "|dp| %.2e" % d[k], "cost %.9g %.9g" % (hip["log"][k].get("final_cost", 0), ref["log"][k].get("final_cost", 0)))
