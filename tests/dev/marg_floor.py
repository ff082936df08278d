"""Measured values behind every gate of tests/test_gpu_marg.py (run on the GPU box): prints the quantity each assert bounds so that
the gates can sit one order of magnitude above the measured floor instead of at a guessed tolerance."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tc-viml_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import orc, synth, tcv as gpu
from util import fro, golden_windows, rel
from test_gpu_marg import marg_batch

pre, main, z = golden_windows()
W, b = marg_batch(gpu, [pre])
P = b.prior(0); d = P.export(); As, bs = P.schur()
JtJ = d["J0"].T @ d["J0"]
print("golden after solve: A' %.2e b' %.2e JtJ %.2e Jtr %.2e x0 %.2e | invariants JtJ~A' %.2e Jtr~b' %.2e" % (
    fro(As, z["marg_A_schur"]), fro(bs, z["marg_b_schur"]), fro(JtJ, z["marg_J0"].T @ z["marg_J0"]), fro(d["J0"].T @ d["r0"], z["marg_J0"].T @ z["marg_r0"]),
    rel(np.concatenate(d["x0"]), z["marg_x0"]), fro(JtJ, As), fro(d["J0"].T @ d["r0"], bs)))
S = np.linalg.eigvalsh(As)
print("   eigenvalues of A': min %.3e max %.3e, below eps: %d" % (S.min(), S.max(), int((S <= 1e-8).sum())))
batch = synth.make_windows(900, 2, frame_shift=-1)
for k in range(2):
    w = synth.window_at(batch, k)
    O = orc.Window(w); O.solve(8, True); st = O.states(); po, dbg = O.marginalize_old()
    w2 = dict(w, pose=st["pose"], speedbias=st["sb"], ex_pose=st["ex"], lam=st["lam"])
    mw = gpu.margin_old_window(w2); Wm = gpu.Window(mw); dr = gpu.margin_old_drops(Wm, mw)
    arr = (gpu._dp * len(dr))(*dr); h = C.c_void_p()
    gpu.check(gpu.lib().tcv_marginalize(Wm.h, arr, len(dr), C.byref(h)))
    Pk = gpu.Prior(h); dk = Pk.export(); Ak, bk = Pk.schur()
    print("standalone at given state %d: A' %.2e b' %.2e JtJ~A'_oracle %.2e Jtr~b'_oracle %.2e" % (k, fro(Ak, dbg["A_schur"]), fro(bk, dbg["b_schur"]), fro(dk["J0"].T @ dk["J0"], dbg["A_schur"]),
                                                                                     fro(dk["J0"].T @ dk["r0"], dbg["b_schur"])))
W, b = marg_batch(gpu, [main])
P = b.prior(0); d = P.export(); As, bs = P.schur()
O = orc.Window(main); O.solve(8, True); po, dbg = O.marginalize_old()
print("incoming prior (after own solve): A' %.2e b' %.2e JtJ %.2e" % (fro(As, dbg["A_schur"]), fro(bs, dbg["b_schur"]), fro(d["J0"].T @ d["J0"], po["J0"].T @ po["J0"])))
# same at bit-identical states
st = O.states()
w2 = dict(main, pose=st["pose"], speedbias=st["sb"], ex_pose=st["ex"], lam=st["lam"])
O2 = orc.Window(w2); po2, dbg2 = O2.marginalize_old()
mw = gpu.margin_old_window(w2); Wm = gpu.Window(mw); dr = gpu.margin_old_drops(Wm, mw)
arr = (gpu._dp * len(dr))(*dr); h = C.c_void_p()
gpu.check(gpu.lib().tcv_marginalize(Wm.h, arr, len(dr), C.byref(h)))
Pk = gpu.Prior(h); dk = Pk.export(); Ak, bk = Pk.schur()
print("incoming prior at identical states: A' %.2e b' %.2e" % (fro(Ak, dbg2["A_schur"]), fro(bk, dbg2["b_schur"])))
# chained solve
W, b = marg_batch(gpu, [pre])
P = b.prior(0); d = P.export(); d["blocks"] = gpu.shifted_prior_blocks(P, W[0])
w = dict(main); w["prior"] = d
Wn = gpu.Window(w)
bn = gpu.Batch([Wn]); bn.solve(gpu.default_options(8, True)); bn.synchronize(); bn.download_states()
s = bn.summaries()[0]
O = orc.Window(main); so = O.solve(8, True)
print("chained solve: cost %.2e pose %.2e sb %.2e" % (abs(s.final_cost - so.final_cost) / so.final_cost, rel(Wn.pose, O.states()["pose"]), rel(Wn.sb, O.states()["sb"])))
