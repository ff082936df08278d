#!/usr/bin/env python3
"""Developer checker (GPU box, host-side ThreadSanitizer): lock-step replays through the native estimator from several host threads with the
instrumented library (`python tc-viml_amd/build.py --tsan`), no torch in the process:

    LD_PRELOAD=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so) TCV_LIB=tc-viml_amd/libtcv_hip_tsan.so \
        TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0" python tests/dev/tsan_replay_drive.py [streams] [host threads] [frames]

What runs concurrently: the threads' tcv_estimators_optimize calls (worker pool sections, plan and camera-half caches, block pool, pinned
staging pool, device memory pool, deferred releases behind stream events), the deferred marginalisation launch handed from one frame's call
to the next (EstInflight), the marginalisation on the thread's second stream, the asynchronous destruction of retired problems."""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd"))
import replay      # noqa: E402
import tcv      # noqa: E402


def main():
    streams = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 14
    n_frames = replay.WINDOW_SIZE + 1 + frames
    seqs = list(replay.EUROC_SEQUENCES)
    st = [replay.simulate_stream_euroc(seqs[s % len(seqs)], n_frames, start_s=0.5 + 2.0 * (s // len(seqs)), max_features=40, max_lines=6, associate=(s % 2 == 0)) for s in range(streams)]
    tcv.check(tcv.lib().tcv_set_device(0))
    ls = [replay.NativeLockstep(st[g::G], num_iterations=8) for g in range(G)]
    counts = [0] * G

    def work(g):
        tcv.check(tcv.lib().tcv_set_device(0))
        for k in range(n_frames):
            counts[g] += ls[g].step(k)

    th = [threading.Thread(target=work, args=(g,)) for g in range(1, G)]
    for t in th:
        t.start()
    work(0)
    for t in th:
        t.join()
    for x in ls:
        x.close()
    print("windows optimised:", sum(counts), "on", G, "host threads,", streams, "streams")
    return 0 if sum(counts) >= streams * (frames - 1) else 1


if __name__ == "__main__":
    sys.exit(main())
