"""Developer tool (GPU): T host threads, each with ONE small batch (W windows) in a loop of solve + gauge fix + marginalisation + synchronize:
calls per second and the kernels' own durations (HIP events of the batch) against the number of threads, cooperative mode on / off.

    python tools/dev_concurrent_small.py [W=1] [reps=60]
"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, ROOT)
import numpy as np
import synth, tcv

Wn = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
TMAX = 8
batch = synth.make_windows(4242, TMAX * Wn)
wins = [synth.window_at(batch, k) for k in range(TMAX * Wn)]


def make(t):
    W = [tcv.Window(w) for w in wins[t * Wn:(t + 1) * Wn]]
    MW = [tcv.margin_old_window(w.win) for w in W]
    M = [tcv.Window(MW[k], share=W[k]) for k in range(Wn)]
    return tcv.Batch(W, M, [tcv.margin_old_drops(W[k], MW[k]) for k in range(Wn)])


for wpw in (0, 1):
    opts = tcv.default_options(8, True, workgroups_per_window=wpw)
    for T in (1, 2, 4, 8):
        bs = [make(t) for t in range(T)]
        out = [None] * T
        bar = threading.Barrier(T)

        def work(t):
            b = bs[t]
            for _ in range(5):
                b.solve(opts); b.gauge_fix(); b.marginalize(); b.synchronize()
            bar.wait()
            t0 = time.perf_counter(); sm = []; mm = []
            for _ in range(reps):
                b.solve(opts); b.gauge_fix(); b.marginalize(); b.synchronize()
                st = b.stats(); sm.append(st["solve_ms"]); mm.append(st["marg_ms"])
            out[t] = (time.perf_counter() - t0, float(np.median(sm)), float(np.median(mm)), float(np.max(sm)))

        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        [x.start() for x in th]; [x.join() for x in th]
        wall = max(o[0] for o in out)
        print("coop %-3s threads %d x %d windows: %7.0f windows/s | per call %.2f ms wall | solve kernel median %.3f ms (max %.3f) marg %.3f ms | workgroups %s" % (
            "on" if wpw == 0 else "off", T, Wn, T * Wn * reps / wall, 1e3 * wall / reps, np.mean([o[1] for o in out]), max(o[3] for o in out), np.mean([o[2] for o in out]),
            bs[0].stats().get("last_solve_workgroups")), flush=True)
        del bs
