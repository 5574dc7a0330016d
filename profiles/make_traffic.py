#!/usr/bin/env python3
"""Per-kernel HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in
separate runs, as MI355X_MICROARCH.md 'HBM' prescribes: the two counters do not fit one pass).

    python profiles/make_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json "workload text"

Units / corrections: both counters are reported in KiB; on gfx950 FETCH_SIZE shows half of the bytes
of wide coalesced reads (guide, re-calibrated in round 1 with rpe_debug_calibrate) -> x2; WRITE_SIZE is
exact for 16-B-per-lane streaming stores.  Values are averages over the launches of a kernel in the
profiled run; bench.py reads the result (pmc_traffic) for the 'traffic' field of the roofline object.
"""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        a = acc[name]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    fetch_csv, write_csv, out, workload = sys.argv[1:5]
    f = per_kernel(fetch_csv, "FETCH_SIZE"); w = per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        if k.startswith("__amd"):
            continue
        fk, fn = f.get(k, (0.0, 0)); wk, wn = w.get(k, (0.0, 0))
        kernels[k] = {"FETCH_SIZE_KB_avg_per_launch": fk, "launches_FETCH_SIZE": fn,
                      "WRITE_SIZE_KB_avg_per_launch": wk, "launches_WRITE_SIZE": wn,
                      "fetch_bytes_corrected": fk * 1024 * 2, "write_bytes": wk * 1024,
                      "hbm_bytes_per_launch": fk * 1024 * 2 + wk * 1024}
    json.dump({"workload": workload,
               "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline",
               "correction": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B; calibrated in round 1 with 4-B and 16-B per-lane "
                             "streaming reads of a known 3.314 GB: 1.657 GB reported); counts L2 misses incl. Infinity-Cache hits",
               "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out, len(kernels), "kernels")


if __name__ == "__main__":
    main()
