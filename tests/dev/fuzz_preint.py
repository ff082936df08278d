#!/usr/bin/env python3
"""Developer checker (GPU): the device IMU pre-integration (IntegrationBase, integration_base.h:13-158) on seeded random sample streams
against the NumPy oracle: 1 .. 3000 samples (a platform at rest for > 10 s is ~2000 at 200 Hz), ragged counts and per-sample dt in ONE
call, dt 1 .. 50 ms, motion from gentle to violent (50 m/s^2, 20 rad/s), large linearisation biases.  Gates of tests/test_gpu_preint.py:
delta_p / delta_v 1e-12, delta_q 1e-13, jacobian / covariance 1e-11 (relative to the largest entry).

    python tests/dev/fuzz_preint.py [cases] [first seed]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import np_oracle as npo      # noqa: E402
import synth      # noqa: E402
import tcv      # noqa: E402
from util import rel      # noqa: E402

NOISE = (synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    worst = dict(p=0.0, v=0.0, q=0.0, J=0.0, P=0.0, dt=0.0)
    bad = []
    for c in range(cases):
        seed = seed0 + c
        rng = np.random.default_rng(700000 + seed)
        nbuf = int(rng.integers(1, 9))
        amp_a, amp_g = float(rng.choice([0.5, 5.0, 50.0])), float(rng.choice([0.1, 2.0, 20.0]))
        counts = [int(rng.choice([1, 2, 7, 70, 400, 3000])) for _ in range(nbuf)]
        first, samples, init, refs = [], [], [], []
        for S in counts:
            dts = rng.choice([0.001, 0.005, 0.02, 0.05], size=S) if rng.random() < 0.5 else np.full(S, float(rng.choice([0.0025, 0.005, 0.01])))
            acc = rng.normal(size=(S + 1, 3)) * amp_a + np.array([0, 0, 9.81]); gyr = rng.normal(size=(S + 1, 3)) * amp_g
            ba = rng.normal(size=3) * float(rng.choice([0.0, 0.02, 0.5])); bg = rng.normal(size=3) * float(rng.choice([0.0, 0.002, 0.1]))
            first.append(len(samples))
            for k in range(S):
                samples.append([dts[k], *acc[k + 1], *gyr[k + 1]])
            init.append([*acc[0], *gyr[0], *ba, *bg])
            refs.append((acc, gyr, dts, ba, bg))
        samples_a = np.ascontiguousarray(np.array(samples, dtype=float).reshape(-1, 7))
        out = (tcv.ImuPreintegration * nbuf)()
        tcv.check(tcv.lib().tcv_preintegrate(nbuf, tcv.iptr(tcv.i32(first)), tcv.iptr(tcv.i32(counts)), tcv.dptr(samples_a), len(samples_a), tcv.dptr(tcv.f64(np.array(init))), tcv.dptr(tcv.f64(NOISE)), out))
        line = []
        for k, (acc, gyr, dts, ba, bg) in enumerate(refs):
            ref = npo.preintegrate(acc, gyr, dts, ba, bg, *NOISE)
            o = out[k]
            e = dict(p=rel(list(o.delta_p), ref["delta_p"]), v=rel(list(o.delta_v), ref["delta_v"]), q=rel(list(o.delta_q), ref["delta_q"]),
                     J=rel(np.array(list(o.jacobian)).reshape(15, 15), ref["jacobian"]), P=rel(np.array(list(o.covariance)).reshape(15, 15), ref["covariance"]),
                     dt=abs(o.sum_dt - ref["sum_dt"]))
            for kk in worst:
                worst[kk] = max(worst[kk], e[kk])
            ok = e["p"] < 1e-12 and e["v"] < 1e-12 and e["q"] < 1e-13 and e["J"] < 1e-11 and e["P"] < 1e-11 and e["dt"] < 1e-12
            if not ok:
                bad.append((seed, k, counts[k], {kk: f"{vv:.1e}" for kk, vv in e.items()}))
            line.append(f"{counts[k]}:{'ok' if ok else 'DIFF'}")
        print(f"case {seed} [acc {amp_a} m/s^2, gyr {amp_g} rad/s]: " + " ".join(line), flush=True)
    print("\nworst:", {k: f"{v:.1e}" for k, v in worst.items()})
    print("flagged:", len(bad))
    for x in bad[:20]:
        print("  ", x)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
