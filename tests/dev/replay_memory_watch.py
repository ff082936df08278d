#!/usr/bin/env python3
"""Developer checker (GPU): device memory the library holds over a long lock-step replay -- live bytes / cached bytes / live buffers every
40 frames (tcv_device_memory_stats): a leak shows as live bytes or buffers growing with the frame count.

    python tests/dev/replay_memory_watch.py [streams] [host threads] [frames]
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
    streams = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 280
    tcv.check(tcv.lib().tcv_set_device(0))
    eng = bench.ReplayEngine(tcv, replay, list(range(streams)), replay.WINDOW_SIZE + 1 + frames, 60, 8, G, 0)
    L = tcv.lib()
    L.tcv_device_memory_stats.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_int)]
    rows = []
    done = 0
    while done < frames - 1:
        step = min(40, frames - 1 - done)
        eng.run(step); done += step
        a, b, n = C.c_ulonglong(), C.c_ulonglong(), C.c_int()
        tcv.check(L.tcv_device_memory_stats(C.byref(a), C.byref(b), C.byref(n)))
        rows.append((done, a.value, b.value, n.value))
        print(f"after {done:4d} frames: live {a.value / 2**20:9.1f} MiB in {n.value:5d} buffers, cached {b.value / 2**20:9.1f} MiB", flush=True)
    # steady state: live bytes and buffers of the second half within 10 % of each other
    half = [r for r in rows if r[0] >= frames // 2]
    lo, hi = min(r[1] for r in half), max(r[1] for r in half)
    nlo, nhi = min(r[3] for r in half), max(r[3] for r in half)
    ok = hi <= 1.1 * lo + 2**20 and nhi <= 1.1 * nlo + 8
    print("second half: live", f"{lo / 2**20:.1f} .. {hi / 2**20:.1f} MiB,", nlo, "..", nhi, "buffers:", "steady" if ok else "GROWING")
    for ls in eng.ls:
        ls.close()
    a, b, n = C.c_ulonglong(), C.c_ulonglong(), C.c_int()
    tcv.check(L.tcv_device_memory_stats(C.byref(a), C.byref(b), C.byref(n)))
    print(f"after closing the estimators: live {a.value / 2**20:.1f} MiB in {n.value} buffers, cached {b.value / 2**20:.1f} MiB")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
