"""tools/ceres_pin/pin_driver's output -> tests/golden/ceres_pin.npz (the reference-executed fixture tests/test_ceres_pin_cpu.py compares the oracle with)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    src = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "ceres_pin.npz")
    d = {"factor_r": [], "factor_J": [], "factor_sizes": [], "iters": [], "pose": [], "speedbias": []}
    for ln in open(src):
        t = ln.split()
        if not t:
            continue
        tag, v = t[0], t[1:]
        if tag == "CERES_VERSION":
            d["ceres_version"] = " ".join(v)
        elif tag == "FACTOR":
            d["factor_sizes"].append([int(x) for x in v[3:]]); d["factor_J"].append([])
        elif tag == "R":
            d["factor_r"].append(np.array(v, dtype=float))
        elif tag[0] == "J" and tag[1:].isdigit():
            d["factor_J"][-1].append(np.array(v, dtype=float))
        elif tag == "SUMMARY":
            d["num_iterations"], d["initial_cost"], d["final_cost"], d["termination"] = int(v[0]), float(v[1]), float(v[2]), int(v[3])
        elif tag == "ITER":
            d["iters"].append([float(x) for x in v])
        elif tag in ("POSE", "SPEEDBIAS"):
            d[tag.lower()].append([float(x) for x in v])
        elif tag == "EX":
            d["ex_pose"] = np.array(v, dtype=float)
        elif tag == "LAM":
            d["lam"] = np.array(v, dtype=float)
    flat = dict(ceres_version=np.array(d.get("ceres_version", "")), num_iterations=d["num_iterations"], initial_cost=d["initial_cost"], final_cost=d["final_cost"],
                termination=d["termination"], iters=np.array(d["iters"]), pose=np.array(d["pose"]), speedbias=np.array(d["speedbias"]), ex_pose=d["ex_pose"], lam=d["lam"],
                n_factors=len(d["factor_r"]))
    for k, (r, Js, sz) in enumerate(zip(d["factor_r"], d["factor_J"], d["factor_sizes"])):
        flat[f"f{k}_r"] = r; flat[f"f{k}_sizes"] = np.array(sz)
        for j, J in enumerate(Js):
            flat[f"f{k}_J{j}"] = J.reshape(len(r), sz[j])
    np.savez_compressed(out, **flat)
    print("wrote", out, "factors", len(d["factor_r"]), "iterations", d["num_iterations"], "final cost", d["final_cost"])


if __name__ == "__main__":
    main()
