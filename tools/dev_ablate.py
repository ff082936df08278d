"""Subtractive profile of the fused solve kernel (developer tool; needs `python tc-viml_amd/build.py --ablate`):

    TCV_LIB=tc-viml_amd/libtcv_hip_abl.so python tools/dev_ablate.py [B]

For every phase bit the benchmark batch is solved with that phase removed (TCV_ABLATE_SKIP); the kernel-time difference to the
full run (every step forced to be accepted, like the skip runs) is what the phase costs under the real overlap conditions."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, ROOT)
import numpy as np
import synth, tcv, bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
names = ["vis_eval", "vis_gather", "lm", "schur", "prior_A", "imu_raw", "imu_whiten", "imu_gather", "prior_B", "fin_scale", "fin_pass", "chain_fwd", "chol",
         "back", "chain_bwd", "lm_back", "dogleg", "plus", "norms", "setup", "ch_T", "ch_owners", "ch_mfma", "ch_fetch"]
batch, wins, keep = bench.build_batches(tcv, synth, 100000, B)
opts = tcv.default_options(8, True)


def run(mask, reps=6):
    os.environ["TCV_ABLATE_SKIP"] = str(mask | (3 << 30))      # bits 31, 30: linear-solver failures ignored, every step accepted (fixed control flow)
    t = []
    for _ in range(reps):
        batch.solve(opts); batch.synchronize(); t.append(batch.stats()["solve_ms"])
    return float(np.median(t[1:]))


base = run(0)
print("B = %d, all phases (every step accepted): %.3f ms" % (B, base), flush=True)
tot = 0.0
for b, nm in enumerate(names):
    if nm == 'setup':
        continue
    t = run(1 << b)
    tot += base - t
    print("  without %-11s %.3f ms   -> phase costs %6.3f ms  (%4.1f %%)" % (nm, t, base - t, 100 * (base - t) / base), flush=True)
print("  sum of the single-phase costs %.3f ms of %.3f" % (tot, base))
groups = {"visual (eval+gather+lm+schur)": 0b1111, "imu (raw+whiten+gather)": 0b11100000, "prior (A+B)": (1 << 4) | (1 << 8), "linearise (all factor families)": 0b111111111,
          "solve (scale+pass+chain+chol+back+bwd+lm_back)": sum(1 << k for k in range(9, 16)), "chain (fwd+bwd)": (1 << 11) | (1 << 14),
          "chain pipelines (T + owners + mfma)": (1 << 20) | (1 << 21) | (1 << 22), "everything": (1 << 19) - 1,
          # the skeleton that is left: what of it the per-solve J0'J0 (bit 19), the copies of the gather programs into LDS (bit 24) and the
          # zeroing of the tiles (bit 25) are
          "everything + per-solve J0'J0": (1 << 20) - 1, "everything + gather-program copies": ((1 << 19) - 1) | (1 << 24),
          "everything + tile zeroing": ((1 << 19) - 1) | (1 << 25), "everything + all three": ((1 << 20) - 1) | (1 << 24) | (1 << 25)}
groups["[experiment] chain over every second step record (6 of 11 steps)"] = 1 << 26
for nm, m in groups.items():
    t = run(m)
    print("  without %-45s %.3f ms  -> %.3f ms (%4.1f %%)" % (nm, t, base - t, 100 * (base - t) / base), flush=True)
