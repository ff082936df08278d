"""Where do the HIP and the oracle replays of V1_03_difficult (association in the loop) part?  Both replays run frame by frame
side by side; after every association the per-observation results (matched map line, errA / errD / overlap, credible flags) are
compared, and the first difference is printed with its margins against the gates of LineCorrespondenceInFrame /
removeLineOutlier / the errD <= dist_th test of OptimizationWithLine.  GPU box: python tests/dev/replay_euroc_flip.py [seq] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tc-viml_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import replay, orc
from replay_oracle import OracleBackend

seq = sys.argv[1] if len(sys.argv) > 1 else "V1_03_difficult"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 200
orc.build()
st = replay.simulate_stream_euroc(seq, F, start_s=0.5, max_features=60, max_lines=8, associate=True)
reps = []
for be in (replay.HipBackend(), OracleBackend()):
    rng = np.random.Generator(np.random.PCG64(0xABCD))
    rp = replay.Replay(be, 8, False)
    rp.map_lines, rp.Rbw, rp.Tbw = st["map_lines"], st["Rbw"], st["Tbw"]
    be.set_map(st["map_lines"], st["Rbw"], st["Tbw"])
    rp.Bas[:] = st["ba"] + rng.normal(size=3) * 0.005; rp.Bgs[:] = st["bg"] + rng.normal(size=3) * 0.0005
    reps.append((rp, rng))
found = False
for k in range(F):
    ready = []
    for rp, rng in reps:
        truth = None
        if k <= replay.WINDOW_SIZE:
            dth = rng.normal(size=3) * 0.005
            truth = (st["gt_p"][k] + rng.normal(size=3) * 0.02, st["gt_R"][k] @ replay.deltaQ_R(dth), st["gt_v"][k] + rng.normal(size=3) * 0.05)
        ready.append(rp.begin_frame(st["imu"][k], st["points"][k], st["lines"][k], truth))
    if not ready[0]:
        continue
    wins = [rp.prepare_window() for rp, _ in reps]
    a, b = reps[0][0], reps[1][0]
    dp = np.abs(a.Ps - b.Ps).max()
    for la, lb in zip(a.linefeatures, b.linefeatures):
        for oa, ob in zip(la["obs"], lb["obs"]):
            same = np.array_equal(oa["world"], ob["world"]) and oa["credible_line"] == ob["credible_line"] and (oa["errD"] > 50.0) == (ob["errD"] > 50.0)
            if not same and not found:
                found = True
                print("frame %d (optimised frame %d), line track %d: first differing association; state difference before it %.2e m" % (k, k - replay.WINDOW_SIZE, la["id"], dp))
                for name, o in (("HIP", oa), ("oracle", ob)):
                    print("  %-6s errA %.7f (gate 0.1745) errD %.5f (gates: dist_th 50) overlap %.7f (gate 0.45) credible %s world %s" % (
                        name, o["errA"], o["errD"], o["overlap"], o["credible_line"], np.array2string(o["world"], precision=3)))
    if [l for l in wins[0]["line"]["frame"]] != [l for l in wins[1]["line"]["frame"]] and found:
        print("frame %d: line factor sets differ (%d vs %d)" % (k, len(wins[0]["line"]["frame"]), len(wins[1]["line"]["frame"])))
    for (rp, _), w in zip(reps, wins):
        rp.apply_result(rp.backend.optimize(w, rp.marg_flag, 8, False))
        rp.finish_frame()
    if found and k > 0 and np.abs(a.Ps - b.Ps).max() > 1e-3:
        print("frame %d: trajectories now %.2e m apart" % (k, np.abs(a.Ps - b.Ps).max()))
        break
if not found:
    print("no differing association in %d frames" % F)
