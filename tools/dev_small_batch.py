"""Developer tool (GPU): kernel latency of SMALL batches -- what a per-sequence replay (BASELINE configs[3] / [4]) and a single
estimator run: B = 1 and B = 5 benchmark windows (cfg 3) and the five ~560-point-factor / ~60-line-factor windows one lock-step frame
of the five EuRoC-trajectory replays hands to the solver.  Prints solve / marginalisation kernel times (HIP events, median of N
repetitions) and, with TCV_LIB=tc-viml_amd/libtcv_hip_prof.so, the per-phase cycle table of the solve kernel.

    python tools/dev_small_batch.py [frame] [reps]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, ROOT)
import numpy as np
import synth, tcv, replay, bench
from tools_common import PHASE_NAMES

_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
FRAME = int(_pos[0]) if len(_pos) > 0 else 60
REPS = int(_pos[1]) if len(_pos) > 1 else 12
PROF = "prof" in os.environ.get("TCV_LIB", "")


class Spy(replay.HipBackend):
    def __init__(self):
        super().__init__()
        self.calls = []

    def optimize(self, win, flag, ni, fi):
        self.calls.append((win, flag))
        return super().optimize(win, flag, ni, fi)


def replay_windows(frame):
    out = []
    for seq in replay.EUROC_SEQUENCES:
        st = replay.simulate_stream_euroc(seq, frame + 12, start_s=0.5, max_features=60, max_lines=8, associate=False)
        spy = Spy()
        replay.run(st, spy, num_iterations=8)
        # the last MARGIN_OLD window of the replay (the common case: a keyframe)
        k = max(i for i, (w, f) in enumerate(spy.calls) if f == replay.MARGIN_OLD)
        out.append(spy.calls[k])
    return out


def make_batch(pairs):
    Ws = [tcv.Window(w) for w, f in pairs]
    Ms, drops = [], []
    for (w, f), W in zip(pairs, Ws):
        mw = tcv.margin_old_window(w); Ms.append(tcv.Window(mw, share=W, prior=W.prior)); drops.append(tcv.margin_old_drops(W, mw))
    return tcv.Batch(Ws, Ms, drops), Ws, Ms


def measure(name, make):
    L = tcv.lib()
    ts, tm = [], []
    prof = np.zeros(32)
    b, Ws, keep = make()
    L.tcv_batch_profile.argtypes = [C.c_void_p, tcv._dp]
    t1 = []
    for rep in range(REPS):
        if PROF and rep == REPS - 1:
            L.tcv_batch_profile(b.h, tcv.dptr(prof))      # clears the accumulators
        b.solve(tcv.default_options(8, True)); b.gauge_fix(); b.marginalize(); b.synchronize()
        s = b.stats(); ts.append(s["solve_ms"]); tm.append(s["marg_ms"])
    co = b.cooperative()
    if co["helpers"] > 0:      # the same plan on one workgroup per window
        for rep in range(REPS):
            b.solve(tcv.default_options(8, True, workgroups_per_window=1)); b.synchronize()
            t1.append(b.stats()["solve_ms"])
    if PROF:
        L.tcv_batch_profile(b.h, tcv.dptr(prof))
    ps = b.plan_stats()
    W0 = Ws[0].plan_stats()
    print("%-44s helpers %d, chunks %d, grid %d | solve %.3f ms (min %.3f)  marg %.3f ms (min %.3f)  sum %.3f ms%s" % (
        name, co["helpers"], co["chunks"], ps["grid"], np.median(ts), min(ts), np.median(tm), min(tm), np.median(ts) + np.median(tm),
        ("  | same plan, one workgroup per window: solve %.3f ms" % np.median(t1)) if t1 else ""), flush=True)
    if PROF and os.environ.get("TCV_DEBUG"):
        print("   marginalisation phases of window 0: see stderr", flush=True)
        b.prior(0)
    if PROF:
        tot = prof.sum()
        print("   phase cycles of the solve kernel (last repetition, summed over the windows of the batch):")
        for n, v in zip(PHASE_NAMES, prof):
            if v > 0:
                print("     %-14s %10.0f  %5.1f %%" % (n, v, 100 * v / tot))


if __name__ == "__main__":
    for B in (() if "--replay-only" in sys.argv else (1, 5)):
        def make(B=B):
            batch, wins, keep = bench.build_batches(tcv, synth, 100000, B)
            return batch, keep[0], keep
        measure("cfg 3 benchmark windows, B = %d" % B, make)
    pairs = replay_windows(FRAME)
    print("replay windows of frame ~%d:" % FRAME, [(len(w["proj"]["frame_i"]), len(w["line"]["frame"]), len(w["lam"])) for w, f in pairs], "(point factors, line factors, landmarks)")
    measure("five replay windows (one lock-step frame)", lambda: make_batch(pairs))
    measure("one replay window", lambda: make_batch(pairs[:1]))
