"""Multi-core CPU baseline: the C restatement (oracle/tcv_oracle.c, one thread per window like Ceres
num_threads = 1) over independent windows of the benchmark workload on N host processes.  Test/measurement infrastructure."""
import os, sys, time
import multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)


def work(args):
    wins, budget = args
    import orc
    orc.lib()
    n, t_solve = 0, 0.0
    t_end = time.perf_counter() + budget
    while time.perf_counter() < t_end:
        ws = [orc.Window(w) for w in wins for _ in range(8)]      # fresh copies (the solve updates the states in place); not timed
        t0 = time.perf_counter()
        for O in ws:
            O.solve(8, True); O.marginalize_old()
        t_solve += time.perf_counter() - t0
        n += len(ws)
    return n, t_solve


if __name__ == "__main__":
    import numpy as np
    import synth, tcv, bench
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else os.cpu_count()
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    batch, wins, _ = bench.build_batches(tcv, synth, 100000, 32)        # windows WITH the GPU-made priors, as in bench.py
    del batch
    with mp.get_context("fork").Pool(nproc) as pool:
        res = pool.map(work, [(wins[(i * 4) % 32:(i * 4) % 32 + 4], budget) for i in range(nproc)])
    tot = sum(r[0] for r in res); tmax = max(r[1] for r in res)
    print({"processes": nproc, "solves": tot, "seconds": round(tmax, 2), "solves_per_s": round(tot / tmax, 1), "per_process": round(tot / tmax / nproc, 1)})
