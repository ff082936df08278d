#!/usr/bin/env python3
"""Developer checker (GPU): host-thread churn -- every frame of a lock-step replay runs on FRESH host threads (bench.ReplayEngine.run(1) spawns
its helper threads per call), so the per-thread stream holder is torn down and its streams re-adopted ~3 x frames times: the streams are
parked without a HIP call, what a thread left in flight is released by the stream's next owner.  Checks: every frame optimised, device memory
steady, nothing live after the estimators are closed.

    python tests/dev/thread_churn.py [streams] [host threads] [frames]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, ROOT)
import bench      # noqa: E402
import replay      # noqa: E402
import tcv      # noqa: E402


def main():
    streams = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    tcv.check(tcv.lib().tcv_set_device(0))
    eng = bench.ReplayEngine(tcv, replay, list(range(streams)), replay.WINDOW_SIZE + 1 + frames, 40, 6, G, 0)
    L = tcv.lib()
    L.tcv_device_memory_stats.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_int)]
    total, rows = 0, []
    for k in range(frames - 1):
        total += eng.run(1)
        if k % 30 == 29:
            a, b, n = C.c_ulonglong(), C.c_ulonglong(), C.c_int()
            tcv.check(L.tcv_device_memory_stats(C.byref(a), C.byref(b), C.byref(n)))
            rows.append((a.value + b.value, n.value))
            print(f"after {k + 1:4d} single-frame calls ({(k + 1) * (G - 1)} host threads come and gone): {total} windows, live + cached {(a.value + b.value) / 2**20:.1f} MiB, {n.value} live buffers", flush=True)
    for ls in eng.ls:
        ls.close()
    a, b, n = C.c_ulonglong(), C.c_ulonglong(), C.c_int()
    tcv.check(L.tcv_device_memory_stats(C.byref(a), C.byref(b), C.byref(n)))
    print(f"after closing: live {a.value} bytes in {n.value} buffers")
    ok = total == streams * (frames - 1) and n.value == 0 and rows[-1][0] <= 1.15 * rows[1][0]
    print("OK" if ok else "FAILED", total, streams * (frames - 1))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
