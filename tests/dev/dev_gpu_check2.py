"""Developer GPU check for the marginalisation kernel."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import synth, tcv, orc, np_oracle as npo

def fro(a, b): return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))
B = 4
pre = synth.make_windows(0, B, frame_shift=-1)
main = synth.make_windows(0, B)
pw = [synth.window_at(pre, k) for k in range(B)]
W = [tcv.Window(w) for w in pw]
MW = [tcv.margin_old_window(w) for w in pw]
M = [tcv.Window(mw, share=W[k]) for k, mw in enumerate(MW)]
drops = [tcv.margin_old_drops(W[k], MW[k]) for k in range(B)]
b = tcv.Batch(W, M, drops)
o = tcv.default_options(8, True)
b.solve(o); b.marginalize(); b.synchronize()
print("stats", b.stats())
b.download_states()
priors = []
for k in range(B):
    P = b.prior(k); d = P.export(); As, bs = P.schur()
    O = orc.Window(pw[k]); O.solve(8, True); po, dbg = O.marginalize_old()
    blocks = tcv.shifted_prior_blocks(P, W[k])
    print(f"win{k}: m {d['m']}/{po['m']} n {d['n']}/{po['n']} blocks_equal {blocks == po['blocks']} "
          f"A_schur rel {fro(As, dbg['A_schur']):.2e} b_schur rel {fro(bs, dbg['b_schur']):.2e} "
          f"J0tJ0 vs As(gpu) {fro(d['J0'].T @ d['J0'], As):.2e} J0tr0 vs bs(gpu) {fro(d['J0'].T @ d['r0'], bs):.2e} "
          f"J0tJ0 vs oracle {fro(d['J0'].T @ d['J0'], po['J0'].T @ po['J0']):.2e} J0tr0 vs oracle {fro(d['J0'].T @ d['r0'], po['J0'].T @ po['r0']):.2e} "
          f"x0 rel {max(fro(a, c) for a, c in zip(d['x0'], po['x0'])):.2e} idx {d['idx'] == po['idx']} sizes {d['sizes'] == po['sizes']}")
    d["blocks"] = blocks
    priors.append((P, d, po))
# standalone marginalisation at the oracle's state: tighter comparison
for k in range(2):
    O = orc.Window(pw[k]); O.solve(8, True); st = O.states(); po, dbg = O.marginalize_old()
    w2 = dict(pw[k]); w2["pose"] = st["pose"]; w2["speedbias"] = st["sb"]; w2["ex_pose"] = st["ex"]; w2["lam"] = st["lam"]
    mw = tcv.margin_old_window(w2)
    Wm = tcv.Window(mw)
    dr = tcv.margin_old_drops(Wm, mw)
    import ctypes as C
    arr = (tcv._dp * len(dr))(*dr)
    h = C.c_void_p()
    tcv.check(tcv.lib().tcv_marginalize(Wm.h, arr, len(dr), C.byref(h)))
    P = tcv.Prior(h); d = P.export(); As, bs = P.schur()
    print(f"standalone win{k}: A_schur rel {fro(As, dbg['A_schur']):.2e} b_schur {fro(bs, dbg['b_schur']):.2e} J0tJ0 {fro(d['J0'].T @ d['J0'], dbg['A_schur']):.2e} J0tr0 {fro(d['J0'].T @ d['r0'], dbg['b_schur']):.2e}")
# chain: main window with the GPU prior vs the oracle chain
wins_g, wins_o = [], []
for k in range(B):
    w = dict(synth.window_at(main, k)); P, d, po = priors[k]
    wg = dict(w); wg["prior"] = dict(d); wo = dict(w); wo["prior"] = po
    wins_g.append(wg); wins_o.append(wo)
Wg = [tcv.Window(w) for w in wins_g]
bg = tcv.Batch(Wg); bg.solve(o); bg.synchronize(); bg.download_states(); s = bg.summaries()
for k in range(B):
    O = orc.Window(wins_o[k]); so = O.solve(8, True); st = O.states(); sg = Wg[k].states()
    print(f"chain win{k}: final {s[k].final_cost:.10g}/{so.final_cost:.10g} state rel pose {fro(sg['pose'], st['pose']):.2e} sb {fro(sg['sb'], st['sb']):.2e} lam {fro(sg['lam'], st['lam']):.2e}")
# timing at batch scale
Bb = 512
big = synth.make_windows(0, Bb, frame_shift=-1)
pw = [synth.window_at(big, k) for k in range(Bb)]
W = [tcv.Window(w) for w in pw]; MW = [tcv.margin_old_window(w) for w in pw]
M = [tcv.Window(mw, share=W[k]) for k, mw in enumerate(MW)]
drops = [tcv.margin_old_drops(W[k], MW[k]) for k in range(Bb)]
b = tcv.Batch(W, M, drops)
b.solve(o); b.marginalize(); b.synchronize()
t = time.time(); b.solve(o); b.marginalize(); b.synchronize(); dt = time.time() - t
print(f"B={Bb}: solve+marg {dt*1e3:.2f} ms, stats {b.stats()}")
