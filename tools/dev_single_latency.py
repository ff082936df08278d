"""Developer tool (GPU): end-to-end latency of the single-window entry points (tcv_solve, tcv_marginalize) incl. packing,
allocation, H2D/D2H -- what a retargeted estimator.cpp pays per frame."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import synth, tcv
from util import golden_windows
pre, main, z = golden_windows()
opts = tcv.default_options(8, True)
for name, w in (("cfg2 (no prior)", pre), ("cfg3 (prior n=75)", main)):
    ts = []
    for rep in range(12):
        W = tcv.Window(w)
        s = tcv.SolverSummary()
        t0 = time.perf_counter()
        tcv.check(tcv.lib().tcv_solve(C.byref(opts), W.h, C.byref(s)))
        ts.append(time.perf_counter() - t0)
    print(name, "tcv_solve ms: first", round(ts[0] * 1e3, 2), "median of rest", round(float(np.median(ts[2:])) * 1e3, 2))
    mw = tcv.margin_old_window(w); ts = []
    for rep in range(8):
        Wm = tcv.Window(mw); dr = tcv.margin_old_drops(Wm, mw); arr = (tcv._dp * len(dr))(*dr); h = C.c_void_p()
        t0 = time.perf_counter()
        tcv.check(tcv.lib().tcv_marginalize(Wm.h, arr, len(dr), C.byref(h)))
        ts.append(time.perf_counter() - t0)
        P = tcv.Prior(h)
    print(name, "tcv_marginalize ms: median", round(float(np.median(ts[1:])) * 1e3, 2))
