"""Developer checker (GPU + oracle): the same simulated stream through the native estimator, the python window management (both on the
HIP back end) and the python window management on the oracle back end, position differences and final costs frame by frame."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tc-viml_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
sys.path.insert(0, ROOT)
import numpy as np
import replay
from test_gpu_replay import OracleBackend
st = replay.simulate_stream_euroc("V2_02_medium", 30, start_s=1.0, max_features=40, max_lines=5, associate=False)
nat = replay.run_many_native([st], num_iterations=8)[0]
py = replay.run_many([st], replay.HipBackend(), num_iterations=8)[0]
ref = replay.run(st, OracleBackend(), num_iterations=8)
for k in range(len(py["t"])):
    ln, lp, lr = nat["log"][k], py["log"][k], ref["log"][k]
    print(k, "flag", lp["flag"], "nlm", lp["n_landmarks"], "nproj", lp["n_proj"], "prior_n", lp["prior_n"],
          "|nat-py| %.2e |py-ref| %.2e |nat-ref| %.2e" % (np.linalg.norm(nat["p"][k]-py["p"][k]), np.linalg.norm(py["p"][k]-ref["p"][k]), np.linalg.norm(nat["p"][k]-ref["p"][k])),
          "cost py %.9g nat %.9g ref %.9g" % (lp.get("final_cost", 0), ln.get("final_cost", 0), lr.get("final_cost", 0)))
