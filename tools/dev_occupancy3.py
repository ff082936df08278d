"""Developer experiment (round-3 review, item 6): would a THIRD chain-layout workgroup per CU pay?

Everything about the solve kernel says latency-bound at two wavefronts per SIMD (VALU-active 0.155 per wave, waiting 0.62), and going from
one to two workgroups per CU gave 1.7x.  The benchmark window does not fit a third of a CU's LDS (53 KB), so the ceiling is measured on a
window that does: the first `--frames` frames of the benchmark windows, no prior (pose system 42 + 1 wide = 6 tiles instead of 15, a 6-step
speed-bias chain), every workgroup limited to 53 KB (TCV_CHAIN_LDS_DOUBLES).  Three configurations, each in a process of its own (the
switches are read once per process):

    A  production build (2 wavefronts / SIMD, 256 registers), 2 workgroups per CU (grid 512)
    B  -DTCV_CHAIN_OCC3 build (3 wavefronts / SIMD, 168 registers: build.py --occ3), 2 workgroups per CU  -> what the register diet costs
    C  the same build, 3 workgroups per CU (grid 768)                                                        -> what the third workgroup buys

    python tc-viml_amd/build.py --occ3 && python tools/dev_occupancy3.py [--frames 6] [--windows 1536]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(frames, B):
    for p in ("tc-viml_amd", "tests"):
        sys.path.insert(0, os.path.join(ROOT, p))
    import numpy as np
    import synth
    import tcv
    from util import sub_window
    batch = synth.make_windows(420000, B)
    wins = [sub_window(synth.window_at(batch, k), frames) for k in range(B)]
    W = [tcv.Window(w) for w in wins]
    b = tcv.Batch(W)
    ps = b.plan_stats()
    st = W[0].plan_stats()
    opts = tcv.default_options(8, True)
    ts = []
    for _ in range(12):
        b.solve(opts); b.synchronize()
        ts.append(b.stats()["solve_ms"])
    ablated = bool(os.environ.get("TCV_ABLATED"))      # tools/dev_phase_split.py: kernels with phases compiled out -- their results are garbage by construction
    if not ablated:
        b.download_states()
    s = b.summaries()
    fin = np.array([s[k].final_cost for k in range(B)]) if not ablated else np.zeros(1)
    print("RESULT %s" % dict(solve_ms=round(float(np.median(ts[3:])), 4), grid=ps["grid"], lds_bytes=ps["lds_bytes"], layout=ps["layout"], n_proj=len(wins[0]["proj"]["frame_i"]),
                             n_line=len(wins[0]["line"]["frame"]), n_vis_chunk=st["n_vis_chunk"], npp=st["npp"], cost_sum=float(fin.sum())), flush=True)


def main():
    if "--worker" in sys.argv:
        i = sys.argv.index("--worker")
        return worker(int(sys.argv[i + 1]), int(sys.argv[i + 2]))
    frames = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 6
    B = int(sys.argv[sys.argv.index("--windows") + 1]) if "--windows" in sys.argv else 1536
    lds = os.environ.get("OCC3_LDS_DOUBLES", "6784")      # 53 KB
    prod = os.path.join(ROOT, "tc-viml_amd", "libtcv_hip.so"); occ3 = os.path.join(ROOT, "tc-viml_amd", "libtcv_hip_occ3.so")
    if not os.path.exists(occ3):
        raise SystemExit("build the three-wavefront variant first: python tc-viml_amd/build.py --occ3")
    rows = []
    for name, lib, grid in (("A: 2 waves/SIMD build (256 VGPRs), 2 workgroups per CU", prod, 512), ("B: 3 waves/SIMD build (168 VGPRs), 2 workgroups per CU", occ3, 512),
                            ("C: 3 waves/SIMD build (168 VGPRs), 3 workgroups per CU", occ3, 768), ("D: 2 waves/SIMD build, grid 768 (the third workgroup waits for a free slot)", prod, 768)):
        env = dict(os.environ, TCV_LIB=lib, TCV_GRID=str(grid), TCV_CHAIN_LDS_DOUBLES=lds)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(frames), str(B)], env=env, capture_output=True, text=True, timeout=900)
        res = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
        if out.returncode != 0 or not res:
            print(name, "FAILED", out.stdout[-400:], out.stderr[-800:])
            continue
        r = eval(res[-1][7:])
        rows.append((name, r))
        print("%-78s solve %.3f ms per %d windows = %.3f us per window  %s" % (name, r["solve_ms"], B, 1e3 * r["solve_ms"] / B, r), flush=True)
    if len(rows) >= 3:
        a, bb, c = rows[0][1]["solve_ms"], rows[1][1]["solve_ms"], rows[2][1]["solve_ms"]
        print("register diet at equal occupancy (B / A): %.3f x;  third workgroup per CU (C / B): %.3f x;  net (C / A): %.3f x" % (bb / a, c / bb, c / a))
        print("same results in every configuration (sum of the final costs):", [r["cost_sum"] for _, r in rows])


if __name__ == "__main__":
    main()
