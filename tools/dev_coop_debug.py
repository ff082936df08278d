"""Developer tool (GPU): launches a cooperative solve and prints the progress marks of group 0's control block while it runs."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import synth, tcv
L = tcv.lib()
wins = [synth.window_at(synth.make_windows(910, 2), k) for k in range(2)]
W = [tcv.Window(w) for w in wins]
print("creating batch", flush=True)
b = tcv.Batch(W)
print("cooperative:", b.cooperative(), b.plan_stats(), flush=True)
out = (C.c_int * 64)()
L.tcv_batch_debug_coop.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
b.solve(tcv.default_options(2, True))
print("launched", flush=True)
for k in range(6):
    time.sleep(0.5)
    rc = L.tcv_batch_debug_coop(b.h, 0, out)
    v = list(out)
    print("t=%.1fs rc %d seq %d cmd %d win %d abort %d done %s | master stage %d seq %d | helpers %s" % (0.5 * (k + 1), rc, v[0], v[1], v[2], v[3], v[8:12], v[16], v[17],
          [(x & 15, x >> 4) for x in v[32:44]]), flush=True)
b.synchronize()
s = b.summaries()
print("done: final costs", [x.final_cost for x in s], "iterations", [x.num_iterations for x in s], "termination", [x.termination for x in s], flush=True)
