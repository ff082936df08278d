#!/usr/bin/env python3
"""Developer checker (GPU box, host-side ThreadSanitizer): bench.py's PCIe-inclusive stream mode -- two to four host threads, each creating,
solving, marginalising and destroying batches of its own on its own library stream, with host-resident and with device-resident priors -- on
the instrumented library (`python tc-viml_amd/build.py --tsan`), no torch in the process.  Invocation as for tsan_replay_drive.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, ROOT)
import bench      # noqa: E402
import synth      # noqa: E402
import tcv      # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    tcv.check(tcv.lib().tcv_set_device(0))
    batch, wins, keep = bench.build_batches(tcv, synth, 100000, B)
    out = bench.stream_figures(tcv, None, keep, B_stream=B // 4, rounds=2, wins=wins)
    print("stream mode driven:", {k: (round(v) if isinstance(v, float) else None) for k, v in out.items() if k.endswith("per_s")})
    return 0


if __name__ == "__main__":
    sys.exit(main())
