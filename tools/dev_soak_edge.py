import os, sys, time
sys.path.insert(0, "tc-viml_amd")
import numpy as np, replay, ate
for name, kw in (("few features", dict(max_features=8)), ("no lines", dict(max_features=30, max_lines=0)), ("many lines", dict(max_features=25, max_lines=12)),
                 ("slow motion", dict(max_features=30, pace=0.97)), ("noisy", dict(max_features=30, pixel_sigma=3.0))):
    try:
        streams = [replay.simulate_stream(5000 + k, 60, **kw) for k in range(3)]
        outs = replay.run_many(streams, replay.HipBackend(), num_iterations=8)
        fl = sum(l["flag"] for o in outs for l in o["log"]); tot = sum(len(o["log"]) for o in outs)
        print(name, {"frames": tot, "second_new": fl, "min_landmarks": min(l["n_landmarks"] for o in outs for l in o["log"]), "max_lines": max(l["n_line"] for o in outs for l in o["log"])})
    except Exception as e:
        print(name, "FAILED", repr(e)[:400])
