"""Round 6, review item 1(b): the bound of a phase-split pipeline, measured before it is built.

Two compile-time ablations of the chain kernel (tcv_solve.hip, -DTCV_ABLATE_CONST: the removed phases' code AND registers are gone) stand in for
the two kernels of a split pipeline:
    lin   factor evaluation + gathers + landmark Schur + IMU + prior -> reduced camera system      (solve phases removed)
    sol   scaling + chain elimination + tiled Cholesky + substitutions + dogleg + Plus               (linearisation phases removed)
each built for 2, 3 and 4 wavefronts per SIMD (256 / 168 / 128 registers: build.py --suffix=lin3 -DTCV_ABLATE_CONST=... -DTCV_CHAIN_OCC=3) and run
with as many workgroups per CU.  The benchmark window does not fit a third of a CU's LDS, so -- like tools/dev_occupancy3.py -- the windows are the
first `--frames` frames of the benchmark windows, every workgroup limited to --lds doubles; 3072 windows per launch (6 / 4 / 3 per workgroup).
Results of the ablated kernels are garbage by construction; every step is forced to be accepted so that the control flow is the benchmark's.

    python tools/dev_phase_split.py [--frames 6] [--windows 3072] [--lds 6400]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIN = sum(1 << k for k in range(9, 19)) | (3 << 30)
SOL = sum(1 << k for k in range(0, 9)) | (1 << 24) | (1 << 25) | (3 << 30)


def main():
    arg = lambda k, d: (sys.argv[sys.argv.index(k) + 1] if k in sys.argv else d)
    frames, B, lds = int(arg("--frames", 6)), int(arg("--windows", 3072)), arg("--lds", "6400")
    lib = lambda s: os.path.join(ROOT, "tc-viml_amd", "libtcv_hip%s.so" % s)
    rows = [("full kernel, production build (256 VGPRs), 2 per CU", lib(""), 512)]
    for occ in (2, 3, 4):
        for kind in ("lin", "sol"):
            rows.append(("%s only, %d waves/SIMD build, %d per CU" % (kind, occ, occ), lib("_%s%d" % (kind, occ)), 256 * occ))
    print("windows: first %d frames of the benchmark windows, no prior; %d per launch; LDS per workgroup %s doubles" % (frames, B, lds))
    res = {}
    for name, path, grid in rows:
        if not os.path.exists(path):
            print("%-60s (library not built)" % name); continue
        env = dict(os.environ, TCV_LIB=path, TCV_GRID=str(grid), TCV_CHAIN_LDS_DOUBLES=lds)
        if path != lib(""):
            env["TCV_ABLATED"] = "1"
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dev_occupancy3.py"), "--worker", str(frames), str(B)], env=env, capture_output=True, text=True, timeout=900)
        r = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
        if out.returncode != 0 or not r:
            print("%-60s FAILED %s" % (name, (out.stdout + out.stderr)[-300:].replace("\n", " | "))); continue
        d = eval(r[-1][7:])
        res[name] = d["solve_ms"]
        print("%-60s %.3f ms per %d windows  (grid %d, lds %d B, chunks %d)" % (name, d["solve_ms"], B, d["grid"], d["lds_bytes"], d["n_vis_chunk"]), flush=True)
    full = res.get(rows[0][0])
    if full:
        for occ in (2, 3, 4):
            a, b = res.get("lin only, %d waves/SIMD build, %d per CU" % (occ, occ)), res.get("sol only, %d waves/SIMD build, %d per CU" % (occ, occ))
            if a and b:
                print("occupancy %d: lin + sol = %.3f ms against the fused kernel's %.3f ms (%.2f x) -- before any hand-over through HBM" % (occ, a + b, full, (a + b) / full))


if __name__ == "__main__":
    main()
