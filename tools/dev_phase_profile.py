"""Developer tool: per-phase cycle breakdown of the solve kernel (needs `python tc-viml_amd/build.py --profile`).
Run with TCV_LIB=tc-viml_amd/libtcv_hip_prof.so."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd"))
import synth, tcv
NAMES = ["setup", "vis_eval", "vis_gather", "lm", "schur", "zero", "imu_raw", "imu_whiten", "imu_gather", "prior", "cost_red",
         "fin_scale", "fin_cauchy", "fin_pass", "chol_diag", "chol_trsm", "chol_upd", "back", "lm_back", "dogleg", "plus", "norms", "other", "chain_fwd", "ch_wait(w0)", "ch_T(wave3)", "ch_owner(w0)", "ch_mfma(w2)", "ch_interval", "ch_wait(w1)", "ch_wait(w2)", "ch_wait(w3)"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
th = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if "--dense" in sys.argv:
    tcv.lib().tcv_set_solver_variant(1)
if "--prior" in sys.argv:       # the bench workload: windows with the n = 75 prior produced by the GPU marginalisation
    sys.path.insert(0, ROOT)
    import bench
    b, wins, keep = bench.build_batches(tcv, synth, 100000, B)
else:
    big = synth.make_windows(0, B)
    W = [tcv.Window(synth.window_at(big, k)) for k in range(B)]
    b = tcv.Batch(W)
o = tcv.default_options(8, True, True, th)
L = tcv.lib(); L.tcv_batch_profile.argtypes = [C.c_void_p, tcv._dp]
b.solve(o); b.synchronize()
out = np.zeros(32); L.tcv_batch_profile(b.h, tcv.dptr(out))
b.solve(o); b.synchronize()
L.tcv_batch_profile(b.h, tcv.dptr(out))
tot = out.sum()
print(b.plan_stats())
print(f"B={B} threads={th} solve_ms={b.stats()['solve_ms']:.3f}; cycles per window-solve (8 it): {tot/B:.0f}")
for n, v in zip(NAMES, out):
    print(f"  {n:12s} {v/B:12.0f} cyc/solve  {100*v/tot:5.1f}%")
