"""Turns rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel-trace only) of bench.py into
profiles/pmc_traffic.json, which bench.py reads for roofline.traffic.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> [out.json]

Units and gfx950 corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KiB
(hbm_bytes = counter * 1024); on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B for wide coalesced reads, so
the read side is reported both raw and doubled (upper bound); WRITE_SIZE is uncalibrated and reported raw."""
import csv, glob, json, os, sys


def per_kernel(dirname, counter):
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"]
            a = acc.setdefault(k, [0.0, set()])
            a[0] += float(row["Counter_Value"])
            a[1].add(row.get("Dispatch_Id"))
    return {k: (v[0], len(v[1])) for k, v in acc.items()}


def main():
    fdir, wdir = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    fetch = per_kernel(fdir, "FETCH_SIZE"); write = per_kernel(wdir, "WRITE_SIZE")
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline`",
           "note": "KiB counters * 1024; FETCH_SIZE x2 is the gfx950 correction for wide coalesced reads (upper bound here)", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        fv, fn = fetch.get(k, (0.0, 0)); wv, wn = write.get(k, (0.0, 0))
        if max(fn, wn) == 0:
            continue
        e = {"launches": max(fn, wn), "fetch_bytes_per_launch_raw": fv * 1024 / max(fn, 1), "write_bytes_per_launch_raw": wv * 1024 / max(wn, 1)}
        e["hbm_bytes_per_launch"] = 2 * e["fetch_bytes_per_launch_raw"] + e["write_bytes_per_launch_raw"]
        res["kernels"][k] = e
        if "solve_kernel" in k:
            res["solve_kernel_hbm_bytes_per_launch"] = e["hbm_bytes_per_launch"]
        if "marg_kernel" in k:
            res["marg_kernel_hbm_bytes_per_launch"] = e["hbm_bytes_per_launch"]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
