"""Developer A/B (GPU): the marginalisation priors of one lock-step frame of replay windows (block mode, chunked projection path) and of
the benchmark windows from two builds of the library, compared bit for bit.

    python tests/dev/marg_ab_libs.py tc-viml_amd/libtcv_hip_prev.so tc-viml_amd/libtcv_hip.so
"""
import os, subprocess, sys, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 3 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
    import numpy as np
    import synth, tcv, bench
    out_file = sys.argv[2]
    sys.argv = sys.argv[:1]      # (dev_small_batch reads its own command line)
    import dev_small_batch as dsb
    out = {}
    pairs = dsb.replay_windows(dsb.FRAME)
    b = dsb.make_batch(pairs)[0]
    b.solve(tcv.default_options(8, True)); b.gauge_fix(); b.marginalize(); b.synchronize(); b.download_priors(compact=True)
    out["replay"] = [p.export() for p in b.priors()]
    batch, wins, keep = bench.build_batches(tcv, synth, 100000, 16)
    batch.solve(tcv.default_options(8, True)); batch.gauge_fix(); batch.marginalize(); batch.synchronize(); batch.download_priors(compact=True)
    out["bench"] = [p.export() for p in batch.priors()]
    t = []
    for _ in range(10):
        b.marginalize(); b.synchronize(); t.append(b.stats()["marg_ms"])
    out["replay_marg_ms"] = float(np.median(t))
    pickle.dump(out, open(out_file, "wb"))
    sys.exit(0)
res = []
for k, lib in enumerate(sys.argv[1:3]):
    f = "/tmp/marg_ab_%d.pkl" % k
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", f], env=dict(os.environ, TCV_LIB=os.path.abspath(lib)))
    res.append(pickle.load(open(f, "rb")))
import numpy as np
for key in ("replay", "bench"):
    same = all(np.array_equal(a["J0"], c["J0"]) and np.array_equal(a["r0"], c["r0"]) and all(np.array_equal(x, y) for x, y in zip(a["x0"], c["x0"]))
               for a, c in zip(res[0][key], res[1][key]))
    print(key, "windows:", len(res[0][key]), "priors bit-identical:", same)
print("marginalisation of the replay frame [ms]:", res[0]["replay_marg_ms"], "->", res[1]["replay_marg_ms"])
