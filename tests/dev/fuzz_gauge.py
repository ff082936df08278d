#!/usr/bin/env python3
"""Developer checker (GPU): the gauge fix (Estimator::double2vector, estimator.cpp:1537-1581) on seeded random windows against the NumPy oracle:
arbitrary attitudes, first poses inside and around the "euler singular point" band (|pitch| within 1 degree of 90, :1549-1555), yaw differences
near +-180 degrees, 1 .. 11 frames.  Gate of tests/test_gpu_gauge.py: 1e-12 (1e-11 on the quaternions written back).

    python tests/dev/fuzz_gauge.py [cases] [first seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import np_oracle as O      # noqa: E402
import tcv      # noqa: E402
from util import rel      # noqa: E402


def rand_R(rng, pitch_deg=None, yaw_deg=None):
    y = rng.uniform(-180, 180) if yaw_deg is None else yaw_deg
    p = rng.uniform(-89, 89) if pitch_deg is None else pitch_deg
    r = rng.uniform(-180, 180)
    return O.ypr2R(np.array([y, p, r]))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    worst = dict(Rs=0.0, Ps=0.0, Vs=0.0, po=0.0)
    bad = []
    kinds = {}
    for c in range(cases):
        seed = seed0 + c
        rng = np.random.default_rng(500000 + seed)
        n = int(rng.integers(1, 12))
        kind = str(rng.choice(["free", "singular R0", "singular pose0", "band edge", "yaw 180"]))
        p0 = p00 = None
        y0 = y00 = None
        if kind == "singular R0":
            p0 = float(rng.choice([-1, 1])) * rng.uniform(89.2, 89.99)
        elif kind == "singular pose0":
            p00 = float(rng.choice([-1, 1])) * rng.uniform(89.2, 89.99)
        elif kind == "band edge":
            p0 = float(rng.choice([-1, 1])) * (89.0 + float(rng.choice([-1e-3, -1e-6, 1e-6, 1e-3])))
        elif kind == "yaw 180":
            y0 = 179.9; y00 = -179.9
        R0 = rand_R(rng, p0, y0); P0 = rng.normal(size=3) * 5
        pose = np.zeros((n, 7)); sb = rng.normal(size=(n, 9))
        for i in range(n):
            R = rand_R(rng, p00 if i == 0 else None, y00 if i == 0 else None)
            q = O.R2q(R); pose[i, 3:] = q / np.linalg.norm(q); pose[i, :3] = rng.normal(size=3) * 5
        ref = O.gauge_fix(R0, P0, pose, sb)
        Rs, Ps, Vs, po = tcv.gauge_fix(R0, P0, pose, sb)
        e = dict(Rs=rel(Rs, ref[0]), Ps=rel(Ps, ref[1]), Vs=rel(Vs, ref[2]), po=rel(po, ref[3]))
        ok = e["Rs"] < 1e-12 and e["Ps"] < 1e-12 and e["Vs"] < 1e-12 and e["po"] < 1e-11
        kinds[(kind, ok)] = kinds.get((kind, ok), 0) + 1
        for k in worst:
            worst[k] = max(worst[k], e[k]) if ok else worst[k]
        if not ok:
            bad.append((seed, kind, n, {k: f"{v:.1e}" for k, v in e.items()}))
    print("by kind:", {f"{a}/{'ok' if b else 'DIFF'}": n for (a, b), n in sorted(kinds.items())})
    print("worst among ok:", {k: f"{v:.1e}" for k, v in worst.items()})
    print("flagged:", len(bad))
    for x in bad[:20]:
        print("  ", x)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
