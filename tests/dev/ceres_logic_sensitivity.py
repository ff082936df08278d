"""Which of the restated Ceres defaults does the solver's convergence behaviour depend on?  (round-2 review, item 7)

The trust-region control logic of ceres::Solve is restated from upstream Ceres 2.x; no Ceres exists in this
image or on the GPU box, so the restatement cannot be run against the real thing.  This script makes the risk checkable from the other
side: the NumPy restatement (oracle/np_oracle.py, `CERES_DEFAULTS`) is run to convergence on the two golden windows
(tests/golden/window.npz) with every default perturbed ONE at a time, and the iterations-to-converge, termination type and final
cost are tabulated next to the SciPy minimum of the same cost function (tests/golden/pins.npz, `sp_nolines_cost`).  A maintainer who
has the reference built against a real Ceres needs ONE number to falsify the restatement: `summary.iterations.size()`
(estimator.cpp:1902 prints it) and `summary.final_cost` on the same window.

    python tests/dev/ceres_logic_sensitivity.py [max_iterations]  > profiles/r03_ceres_logic_sensitivity.txt

CPU only; imports the oracle (test infrastructure), never the product."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("oracle", "tests", os.path.join("tests", "golden")):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import np_oracle as npo
from util import golden_windows, load
from make_golden_pins import no_lines

MAXIT = int(sys.argv[1]) if len(sys.argv) > 1 else 300
pre, main, z = golden_windows()
P = load("pins.npz")
cmin = float(P["sp_nolines_cost"])
CASES = [("defaults (upstream Ceres 2.x as restated)", {})]
for k, vals in (("jacobi_scaling", ["off", "1/norm"]),
                ("min_lm_diagonal", [1e-9, 1e-3, 1.0]),
                ("min_mu", [1e-12, 1e-10, 1e-6, 1e-4]),
                ("mu_decrease", ["keep"]),
                ("mu_increase_factor", [2.0, 100.0]),
                ("initial_trust_region_radius", [1e0, 1e2, 1e6, 1e16]),
                ("decrease_threshold", [0.1, 0.5]),
                ("increase_threshold", [0.5, 0.9]),
                ("radius_increase_factor", [2.0, 10.0]),
                ("min_relative_decrease", [0.0, 1e-1]),
                ("function_tolerance", [1e-5, 1e-7, 1e-8, 1e-10]),
                ("parameter_tolerance", [1e-6, 1e-10])):
    for v in vals:
        CASES.append(("%s = %s (default %s)" % (k, v, npo.CERES_DEFAULTS[k]), {k: v}))


def run(win, over):
    prob = npo.Problem(win)
    t0 = time.time()
    x, s = npo.solve(prob, MAXIT, False, ceres_defaults=over)
    its = s["iterations"]
    n_ok = sum(1 for r in its[1:] if r.get("step_ok"))
    n_rej = sum(1 for r in its[1:] if not r.get("step_ok") and not r.get("invalid"))
    n_inv = sum(1 for r in its[1:] if r.get("invalid"))
    cases = [r.get("case", 0) for r in its[1:]]
    c8 = its[min(8, len(its) - 1)]["cost"]
    return dict(n=len(its), ok=n_ok, rej=n_rej, inv=n_inv, term=s["termination"], cost=s["final_cost"], c8=c8,
                cases="".join(str(c) for c in cases[:12]), sec=time.time() - t0)


print("Sensitivity of the restated trust-region loop to each Ceres default (NumPy oracle, run to convergence, max %d iterations)" % MAXIT)
print("window A: golden main window WITHOUT line factors (cfg 2 + prior n = 75): SciPy minimum of the same cost = %.12f" % cmin)
print("window B: golden main window with its 40 line factors (cfg 3; the line Jacobian is not a derivative, so no minimiser pin exists)")
print("columns: summary.iterations.size() (incl. iteration 0) | accepted / rejected / invalid | termination | final cost | excess over the")
print("SciPy minimum (A only) | cost after the benchmark's 8 iterations | dogleg cases of the first 12 iterations (1 GN, 2 Cauchy, 3 interpolated)")
for name, win in (("A", no_lines(main)), ("B", main)):
    print("\n== window %s ==" % name)
    base = None
    for label, over in CASES:
        r = run(win, over)
        if base is None:
            base = r
        ex = "%+.3f %%" % (100.0 * (r["cost"] / cmin - 1.0)) if name == "A" else "   n/a "
        flag = "" if (r["n"] == base["n"] and abs(r["cost"] - base["cost"]) <= 1e-9 * base["cost"]) else ("  <-- differs" if abs(r["n"] - base["n"]) > 2 or abs(r["cost"] - base["cost"]) > 1e-4 * base["cost"] else "  (close)")
        print("%-58s %4d | %3d /%3d /%2d | %-22s | %.9f | %9s | %.6f | %s%s" % (label, r["n"], r["ok"], r["rej"], r["inv"], r["term"], r["cost"], ex, r["c8"], r["cases"], flag), flush=True)
