"""Developer tool: where do the wavefronts of the chain kernel's workgroups sit (SIMD, wave slot), and what does re-placing the wavefront
roles cost / gain (TCV_ROLE_MODE, tcv_solve.hip: solve_kernel)?   python tools/dev_role_modes.py [B]"""
import os, sys, subprocess, json, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, ROOT)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
if "--hist" in sys.argv:
    import synth, tcv, bench
    b, wins, keep = bench.build_batches(tcv, synth, 100000, B)
    o = tcv.default_options(8, True, True, 256)
    L = tcv.lib(); L.tcv_batch_profile.argtypes = [C.c_void_p, tcv._dp]
    out = np.zeros(32); L.tcv_batch_profile(b.h, tcv.dptr(out))
    b.solve(o); b.synchronize()
    L.tcv_batch_profile(b.h, tcv.dptr(out))
    print("mode", os.environ.get("TCV_ROLE_MODE"), "workgroup launches:", out[24] + out[25], "of them with four distinct SIMDs:", out[25])
    print("  wave x SIMD:", out[:16].reshape(4, 4).astype(int).tolist())
    print("  wave slot of wave 0:", out[16:24].astype(int).tolist())
    sys.exit(0)
env = dict(os.environ)
env["TCV_ROLE_MODE"] = "16"
subprocess.call([sys.executable, __file__, str(B), "--hist"], env=env)
for rep in range(2):
    for mode in (0, 1, 2, 3, 4, 5, 6):
        env["TCV_ROLE_MODE"] = str(mode)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print("TCV_ROLE_MODE=%d  solve %.3f ms  marg %.3f ms  %.1f K solves/s" % (mode, d["kernel_ms"]["solve"], d["kernel_ms"]["marginalize"], d["value"] / 1e3), flush=True)
        except Exception as e:
            print("mode", mode, "failed", e, r.stderr[-500:])
