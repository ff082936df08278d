"""One run of the benchmark batch with the phases in MASK removed (ablation build), for a profiler to wrap:

    TCV_LIB=tc-viml_amd/libtcv_hip_abl.so python3 tools/dev_ablate_one.py MASK [B] [reps]

(tools/dev_ablate_pmc.sh: hardware counters per removed phase -> which phase the LDS bank conflicts, SALU instructions ... belong to)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, ROOT)
import synth, tcv, bench

mask = int(sys.argv[1], 0)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
batch, wins, keep = bench.build_batches(tcv, synth, 100000, B)
opts = tcv.default_options(8, True)
os.environ["TCV_ABLATE_SKIP"] = str(mask | (3 << 30))
for _ in range(reps):
    batch.solve(opts); batch.synchronize()
print("mask %#x solve_ms %.3f" % (mask, batch.stats()["solve_ms"]))
