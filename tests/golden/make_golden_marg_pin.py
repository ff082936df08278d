"""Independent pin for the marginalisation's Schur step (run in the authoring container; writes tests/golden/marg_pin.npz):

    python tests/golden/make_golden_marg_pin.py

The product forms A' = Arr - Arm Amm^+ Amr by one of two routes (Cholesky factor of Amm when its rank is proven, eigen-decomposition
otherwise); both oracles use the eigen-decomposition like the reference (marginalization_factor.cpp:262-282).  This fixture
is the same quantity at 50 digits: every factor of the MARGIN_OLD set of the two golden windows is evaluated by the NumPy restatement at the
window's initial states (binary64 r, J: exact inputs), A = sum J'J and b = sum J'r are accumulated with mpmath, Amm is inverted by 50-digit
LU (its smallest eigenvalue is far above eps = 1e-8 for these windows, asserted below, so the pseudo-inverse is the inverse) and A', b' are
rounded to binary64 once at the end.  What remains between an FP64 implementation and this pin is the rounding of its own J, r and sums --
the reproducibility floor of A' (a difference of 1e14-sized terms) -- not the route it takes through Amm."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tc-viml_amd"), os.path.join(ROOT, "tests"), HERE):
    sys.path.insert(0, p)
import np_oracle as npo
from util import golden_windows


def mp_margin_old(win, dps=50):
    import mpmath as mp
    mp.mp.dps = dps
    F = mp.mpf
    prob = npo.Problem(win)
    x = prob.x0()
    # the factor set and the [m | n] order exactly as np_oracle.marginalize_old / _marginalize build them
    pri, dbg = npo.marginalize_old(prob, x)
    m, n = dbg["m"], dbg["n"]
    order = {nm_i: k for k, nm_i in enumerate((nm, i) for (nm, i, g) in prob.blocks)}
    gsize = {(nm, i): g for (nm, i, g) in prob.blocks}
    idx, pos = {}, 0
    for b in dbg["drop"] + dbg["keep"]:
        idx[b] = pos
        pos += 6 if gsize[b] == 7 else gsize[b]
    assert pos == m + n
    A = mp.zeros(pos, pos)
    bv = [F(0)] * pos
    w = prob.win
    for fac in prob.factors():
        kind, k, blks = fac
        take = (kind == "prior") or (kind == "imu" and blks[0] == ("pose", 0) and float(w["imu"]["sum_dt"][k]) < 10.0) or \
               (kind in ("proj", "proj_td") and blks[0] == ("pose", 0))
        if not take:
            continue
        r, Js, _ = prob.eval_factor(fac, x, True, None)
        cols = []
        for i, b in enumerate(blks):
            si = 6 if gsize[b] == 7 else gsize[b]
            for c in range(si):
                cols.append((idx[b] + c, Js[i][:, c]))
        for a, ca in cols:
            for row in range(len(r)):
                va = F(float(ca[row]))
                if va == 0:
                    continue
                bv[a] += va * F(float(r[row]))
                for b2, cb in cols:
                    if b2 <= a and cb[row] != 0.0:
                        A[a, b2] += va * F(float(cb[row]))
    for a in range(pos):
        for b2 in range(a):
            A[b2, a] = A[a, b2]
    Amm = A[:m, :m]
    lam_min = min(mp.eigsy(Amm, eigvals_only=True))
    assert lam_min > 1e-6, lam_min
    Ainv = mp.inverse(Amm)
    Amr, Arm, Arr = A[:m, m:], A[m:, :m], A[m:, m:]
    A2 = Arr - Arm * Ainv * Amr
    bm, br = mp.matrix(bv[:m]), mp.matrix(bv[m:])
    b2v = br - Arm * (Ainv * bm)
    A2f = np.array([[float(A2[i, j]) for j in range(n)] for i in range(n)])
    b2f = np.array([float(b2v[i]) for i in range(n)])
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    print("m %d n %d lambda_min(Amm) %.3e; float64 NumPy oracle vs 50 digits: A' %.2e b' %.2e" % (m, n, float(lam_min), rel(dbg["A_schur"], A2f), rel(dbg["b_schur"], b2f)), flush=True)
    return A2f, b2f


def main():
    pre, main_w, z = golden_windows()
    out = {}
    for w, p in ((pre, "pre_"), (main_w, "main_")):
        A2, b2 = mp_margin_old(w)
        out["mg_" + p + "A"] = A2; out["mg_" + p + "b"] = b2
    np.savez_compressed(os.path.join(HERE, "marg_pin.npz"), **out)
    print("wrote marg_pin.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
