"""Developer timing (GPU): where a frame of the no-wait prior hand-over spends its host time (B windows)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, ROOT)
import numpy as np
import synth, tcv
T0 = time.perf_counter()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 300
nowait = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
pre = synth.make_windows(9400, B, frame_shift=-1); print('make 1 %.2f' % (time.perf_counter() - T0), flush=True)
pw = [synth.window_at(pre, k) for k in range(B)]; print('window_at 1 %.2f' % (time.perf_counter() - T0), flush=True)
main = synth.make_windows(9400, B)
mw = [synth.window_at(main, k) for k in range(B)]; print('second %.2f' % (time.perf_counter() - T0), flush=True)
opts = tcv.default_options(8, True)
def marg_batch(wins, Ws):
    MW = [tcv.margin_old_window(W.win) for W in Ws]
    M = [tcv.Window(m, share=Ws[k], prior=Ws[k].prior) for k, m in enumerate(MW)]
    return tcv.Batch(Ws, M, [tcv.margin_old_drops(Ws[k], MW[k]) for k in range(len(wins))])
print('setup: data %.2f s' % (time.perf_counter() - T0), flush=True)
W = [tcv.Window(w) for w in pw]
b = marg_batch(pw, W)
old = []
print('setup: first batch %.2f s' % (time.perf_counter() - T0), flush=True)
for frame in range(3):
    t = [time.perf_counter()]
    b.solve(opts); b.gauge_fix(); b.download_states(); s = b.summaries(); t.append(time.perf_counter())
    b.marginalize(); t.append(time.perf_counter())
    pri = b.priors_device(nowait=nowait); t.append(time.perf_counter())
    blocks = [tcv.shifted_prior_blocks(pri[k], W[k]) for k in range(B)]; t.append(time.perf_counter())
    old.append(b)
    W = [tcv.Window(dict(mw[k], prior=dict(blocks=blocks[k])), prior=pri[k]) for k in range(B)]; t.append(time.perf_counter())
    b = marg_batch(mw, W); t.append(time.perf_counter())
    print("frame", frame, ["%.3f" % (t[i + 1] - t[i]) for i in range(len(t) - 1)], "solve+dl | marg launch | handles | blocks | windows | batch", flush=True)

t0 = time.perf_counter()
for ob in old:
    st = ob.marg_status()
print("statuses %.3f s" % (time.perf_counter() - t0), flush=True)
t0 = time.perf_counter(); ex = [p.export() for p in pri]; print("exports %.3f s" % (time.perf_counter() - t0), flush=True)
t0 = time.perf_counter(); del pri, ex; print("del priors %.3f s" % (time.perf_counter() - t0), flush=True)
t0 = time.perf_counter(); del old; print("del old batches %.3f s" % (time.perf_counter() - t0), flush=True)
t0 = time.perf_counter(); del b, W; print("del last batch %.3f s" % (time.perf_counter() - t0), flush=True)
print("total %.2f s" % (time.perf_counter() - T0), flush=True)
