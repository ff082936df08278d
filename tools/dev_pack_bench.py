#!/usr/bin/env python3
"""Throughput of the packer (tc-viml_amd/csrc/tcv_pack.cpp: plans + data sizes, what tcv_batch_create's first pass does) on windows shaped
like a live estimator's -- no device needed.  Replays a short EuRoC-trajectory stream through the Python window management with a back end
that only records the windows, then packs variants of them on 1 .. N host threads (tcv_problems_pack_bench).

    python tools/dev_pack_bench.py [threads ...]          TCV_DEBUG_PACK2=1: per-phase times
    PACK_BENCH_LAND=150 PACK_BENCH_LINES=88 PACK_BENCH_DROP=1: ~560 point + 88 line factors per window, like a 60-feature replay frame
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd"))
import replay   # noqa: E402
import synth    # noqa: E402
import tcv      # noqa: E402


def windows(n_land=int(os.environ.get("PACK_BENCH_LAND", "90")), reps=160):
    """replay-sized windows with distinct structures: ragged tracks drawn per window"""
    rng = np.random.Generator(np.random.PCG64(11))
    out = []
    for r in range(reps):
        w = dict(synth.window_at(synth.make_windows(7000 + r, 1, n_landmarks=n_land + int(rng.integers(-10, 10)), n_lines=int(os.environ.get("PACK_BENCH_LINES", "40"))), 0))
        pr = {k: np.asarray(v) for k, v in w["proj"].items()}
        keep = np.ones(len(pr["landmark"]), bool)
        for l in range(int(pr["landmark"].max()) + 1):
            idx = np.nonzero(pr["landmark"] == l)[0]
            keep[idx[int(rng.integers(max(1, len(idx) - int(os.environ.get("PACK_BENCH_DROP", "99"))), len(idx) + 1)):]] = False
        w["proj"] = {k: (v[keep] if isinstance(v, np.ndarray) and v.shape[:1] == keep.shape else v) for k, v in pr.items()}
        out.append(w)
    return out


def main():
    threads = [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16]
    L = tcv.lib()
    L.tcv_problems_pack_bench.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    for nt in threads:
        Ws = [tcv.Window(w) for w in windows(reps=int(os.environ.get("PACK_BENCH_WINDOWS", "320")))]      # fresh structures per measurement: every plan is built
        arr = (C.c_void_p * len(Ws))(*[w.h for w in Ws])
        s = C.c_double()
        tcv.check(L.tcv_problems_pack_bench(arr, len(Ws), nt, 4, C.byref(s)))
        st = Ws[0].plan_stats()
        print(f"threads {nt:2d}: {1e6 * s.value / len(Ws):7.1f} us wall per window, {1e6 * s.value / len(Ws) * nt:7.1f} us core time per window "
              f"({len(Ws)} windows, first: {st['nland']} landmarks, plan {st['plan_ints'] * 4 / 1024:.0f} KB)")
        del Ws
    st = (C.c_longlong * 4)()
    L.tcv_plan_cache_stats(st)
    print("whole-plan cache hits / misses, camera-half hits / misses:", list(st))


if __name__ == "__main__":
    main()
