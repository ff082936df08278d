"""Developer tool (GPU): block-mode marginalisation, chunked path vs the factor-by-factor path (TCV_MARG_BLOCK_SERIAL=1): where A' / b' differ."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tc-viml_amd"))
import numpy as np
import synth, tcv
np.set_printoptions(linewidth=250, precision=2)
w = synth.window_at(synth.make_windows(902, 1, n_landmarks=int(sys.argv[1]) if len(sys.argv) > 1 else 400, frame_shift=-1), 0)
mw = tcv.margin_old_window(w)
res = {}
for mode in ("serial", "chunked"):
    if mode == "serial":
        os.environ["TCV_MARG_BLOCK_SERIAL"] = "1"
    else:
        os.environ.pop("TCV_MARG_BLOCK_SERIAL", None)
    Wm = tcv.Window(mw)
    dr = tcv.margin_old_drops(Wm, mw)
    arr = (tcv._dp * len(dr))(*dr)
    h = C.c_void_p()
    tcv.check(tcv.lib().tcv_marginalize(Wm.h, arr, len(dr), C.byref(h)))
    P = tcv.Prior(h)
    res[mode] = P.schur()
    print(mode, "dims", P.dims())
As, bs = res["serial"]; Ac, bc = res["chunked"]
D = np.abs(As - Ac)
print("max |A'| %.3e  max diff %.3e  rel fro %.3e   b' rel %.3e" % (np.abs(As).max(), D.max(), np.linalg.norm(As - Ac) / np.linalg.norm(As), np.linalg.norm(bs - bc) / np.linalg.norm(bs)))
n = As.shape[0]
nb = (n + 5) // 6
B = np.zeros((nb, nb))
for i in range(nb):
    for j in range(nb):
        B[i, j] = D[6 * i:6 * i + 6, 6 * j:6 * j + 6].max() / max(np.abs(As[6 * i:6 * i + 6, 6 * j:6 * j + 6]).max(), 1e-300)
print("relative difference per 6 x 6 block:")
print(B)
