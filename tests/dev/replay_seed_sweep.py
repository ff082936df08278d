"""Developer checker (GPU + oracle): HIP vs oracle back end over several simulated association replays (seeds), max position
difference and whether every association decision agrees.  TCV_MARG_EIG_MM=1 / TCV_MARG_NT select kernel variants."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tc-viml_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import replay
from replay_oracle import OracleBackend
seeds = [int(a) for a in sys.argv[1:]] or list(range(1, 9))
for seed in seeds:
    for assoc in (True, False):
        stream = replay.simulate_stream(seed, 30, max_features=30, associate=assoc)
        hip = replay.run(stream, replay.HipBackend(), num_iterations=8)
        ref = replay.run(stream, OracleBackend(), num_iterations=8)
        d = np.linalg.norm(hip["p"] - ref["p"], axis=1)
        same = [l["n_line"] for l in hip["log"]] == [l["n_line"] for l in ref["log"]]
        print("seed", seed, "associate", assoc, "max |dp| %.2e  first frame above 1e-4: %s  decisions agree: %s" % (d.max(), (np.nonzero(d > 1e-4)[0][:1].tolist() or None), same), flush=True)
