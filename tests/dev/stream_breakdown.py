"""Developer tool (GPU): where the time of one PCIe-inclusive pass goes (tcv_batch_create: pack / alloc + upload / marg attach; compute;
download) for a 512-window batch of benchmark windows.  TCV_DEBUG_PACK=1 prints the split of tcv_batch_create."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, ROOT)
import numpy as np
import synth, tcv, bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
batch, wins, keep = bench.build_batches(tcv, synth, 100000, B)
Wm, Mm, dropsm = keep[:3]
opts = tcv.default_options(8, True)
for rep in range(4):
    t0 = time.perf_counter()
    b = tcv.Batch(Wm, Mm, dropsm)
    t1 = time.perf_counter()
    b.solve(opts); b.gauge_fix(); b.marginalize(); b.synchronize()
    t2 = time.perf_counter()
    b.download_states()
    t3 = time.perf_counter()
    b.download_priors(compact=True)
    t4 = time.perf_counter()
    for k in range(B):
        b.prior(k)
    t5 = time.perf_counter()
    del b
    t6 = time.perf_counter()
    print("rep %d: create %.2f ms, compute %.2f, states D2H %.2f, priors D2H %.2f, prior objects %.2f, destroy %.2f" % (rep, *(1e3 * (b_ - a_) for a_, b_ in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6)))), flush=True)
# round 4: the same pass with the priors DEVICE-RESIDENT (tcv_batch_get_priors_device: nothing of them is packed, uploaded or downloaded)
import ctypes as C
pdev = keep[3]
n = B
tcv.check(tcv.lib().tcv_problems_set_marginalization_prior((C.c_void_p * n)(*[w.h for w in Wm[:n]]), (C.c_void_p * n)(*[p.h for p in pdev[:n]]), n))
tcv.check(tcv.lib().tcv_problems_set_marginalization_prior((C.c_void_p * n)(*[w.h for w in Mm[:n]]), (C.c_void_p * n)(*[p.h for p in pdev[:n]]), n))
for rep in range(4):
    t0 = time.perf_counter()
    b = tcv.Batch(Wm, Mm, dropsm)
    t1 = time.perf_counter()
    b.solve(opts); b.gauge_fix(); b.marginalize(); b.synchronize()
    t2 = time.perf_counter()
    b.download_states()
    t3 = time.perf_counter()
    raw = b.priors_device_raw()
    t4 = time.perf_counter()
    tcv.lib().tcv_priors_destroy(raw, len(raw))
    del b
    t5 = time.perf_counter()
    print("device-resident priors, rep %d: create %.2f ms, compute %.2f, states D2H %.2f, prior handles %.2f, destroy %.2f" % (rep, *(1e3 * (b_ - a_) for a_, b_ in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5)))), flush=True)
