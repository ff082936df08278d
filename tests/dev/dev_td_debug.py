"""first dogleg step of an ESTIMATE_TD window, HIP vs NumPy oracle, block by block (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tc-viml_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, synth, tcv, np_oracle as NO
w = synth.with_time_offset(synth.window_at(synth.make_windows(21, 1), 0), 21, TR=0.0)
W = tcv.Window(w); b = tcv.Batch([W])
b.solve(tcv.default_options(1, True, True, 256, True)); b.synchronize()
d = b.first_step(0)
s = b.summaries()[0]
P = NO.Problem(w)
x, so = NO.solve(P, 1, True)
do = so["iterations"][1]["delta"]
print("cost0", s.cost[0], so["iterations"][0]["cost"], "cost1", s.cost[1], so["iterations"][1]["cost"], "case", s.dogleg_case[1], so["iterations"][1]["case"])
print("len", len(d), len(do))
off = 0
for (nm, i, g) in P.blocks:
    lo = P.loff[(nm, i)]
    if lo < 0: continue
    ls = 6 if g == 7 else g
    a, c = d[lo:lo + ls], do[lo:lo + ls]
    e = np.abs(a - c).max() / max(1e-12, np.abs(c).max())
    if e > 1e-6 and nm != "lam": print(nm, i, "rel err %.2e" % e, a[:3], c[:3])
lam_err = np.abs(d[P.nc:] - do[P.nc:]).max() / np.abs(do[P.nc:]).max()
print("lam rel err %.2e" % lam_err)
