#!/usr/bin/env python3
"""Developer checker (GPU): ESTIMATE_TD windows (every point factor a ProjectionTdFactor on para_Td[0], estimator.cpp:1703-1707, :1757-1763)
on the randomised structures of tests/dev/fuzz_solve.py, chain and dense layout, against the NumPy oracle (oracle/np_oracle.py is the checker
here, as in tests/test_gpu_td.py).  Ill-posed mutations are priced like in fuzz_solve.py: by what the oracle itself moves under 1e-13 input noise.

    python tests/dev/fuzz_td.py [cases] [first seed]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fuzz_solve as fz      # noqa: E402
import np_oracle as NO      # noqa: E402
from util import rel      # noqa: E402

tcv, synth = fz.tcv, fz.synth


def oracle(w):
    P = NO.Problem(w)
    x, so = NO.solve(P, 8, True)
    return x, so


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    tally, bad = {}, []
    for c in range(cases):
        seed = seed0 + c
        rng = np.random.Generator(np.random.PCG64(seed))
        w_hip, w_orc, exc, note = fz.make_case(rng, seed)
        if exc or len(w_hip["proj"]["landmark"]) == 0:      # (the NumPy oracle has no constant-extrinsic switch; a window without point factors has no Td factor)
            continue
        TR = float(rng.choice([0.0, 0.02]))
        w_hip = synth.with_time_offset(w_hip, seed, TR=TR); w_orc = synth.with_time_offset(w_orc, seed, TR=TR)
        try:
            x, so = oracle(w_orc)
        except Exception as e:      # noqa: BLE001
            print(f"case {seed} [{note}]: oracle refused: {str(e)[:100]}"); continue
        sens = dict(cost=0.0, pose=0.0, sb=0.0, lam=0.0, td=0.0)
        for rep in range(2):
            r2 = np.random.Generator(np.random.PCG64(811 + rep))
            w2 = dict(w_orc)
            for key in ("lam", "pose", "speedbias"):
                a = np.asarray(w_orc[key], dtype=float)
                w2[key] = a * (1 + 1e-13 * r2.standard_normal(a.shape))
            x2, so2 = oracle(w2)
            sens["cost"] = max(sens["cost"], abs(so2["final_cost"] - so["final_cost"]) / max(so["final_cost"], 1e-12))
            for key in ("pose", "sb", "lam"):
                sens[key] = max(sens[key], rel(x2[key], x[key]))
            sens["td"] = max(sens["td"], abs(x2["td"][0] - x["td"][0]))
        row = []
        for lay in ("chain", "dense"):
            tcv.check(tcv.lib().tcv_set_solver_variant(0 if lay == "chain" else 1))
            try:
                W = tcv.Window(w_hip); b = tcv.Batch([W])
                b.solve(tcv.default_options(8, True)); b.synchronize(); b.download_states()
                s = b.summaries()[0]
                its = so["iterations"]
                soft = []
                verdict = "ok"
                if s.num_iterations != len(its) or [s.step_ok[i] for i in range(1, len(its))] != [int(it["step_ok"]) for it in its[1:]]:
                    verdict = "trace"
                else:
                    checks = [("cost", abs(s.final_cost - so["final_cost"]) / max(so["final_cost"], 1e-12), 1e-6), ("pose", rel(W.pose, x["pose"]), 1e-6), ("sb", rel(W.sb, x["sb"]), 1e-6),
                              ("lam", rel(W.lam, x["lam"]), 1e-5), ("td", abs(W.td[0] - x["td"][0]), 1e-6 * max(1e-3, abs(x["td"][0])))]
                    for name, d, tol in checks:
                        if not d < tol:
                            if d < 30 * sens[name]:
                                soft.append(f"{name} {d:.1e} / oracle {sens[name]:.1e}")
                            else:
                                verdict = "DIFF"; soft.append(f"{name} {d:.2e} (oracle moves {sens[name]:.1e})")
                    if verdict == "ok" and soft:
                        verdict = "ok~"
                detail = "; ".join(soft)
                layout = b.plan_stats()["layout"]
            except Exception as e:      # noqa: BLE001
                verdict, detail, layout = "ERROR", str(e)[:120], "?"
            finally:
                tcv.check(tcv.lib().tcv_set_solver_variant(0))
            tally[(lay, verdict)] = tally.get((lay, verdict), 0) + 1
            row.append(f"{lay}:{verdict}" + (f"({detail})" if detail else "") + f"[{layout}]")
            if verdict in ("DIFF", "ERROR"):
                bad.append((seed, note, lay, verdict, detail))
        print(f"case {seed} [{note}; TR {TR}; {len(w_hip['proj']['landmark'])} point, {len(w_hip['line']['frame'])} line factors, {len(w_hip['lam'])} landmarks, prior {'yes' if w_hip.get('prior') is not None else 'no'}]: " + "  ".join(row), flush=True)
    print("\ntally:", {f"{a}/{b}": n for (a, b), n in sorted(tally.items())})
    print("flagged:", len(bad))
    for xx in bad:
        print("  ", xx)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
