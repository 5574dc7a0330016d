#!/usr/bin/env python3
"""Per-kernel counters of one workload from separate rocprofv3 passes (MI355X_MICROARCH.md 'rocprofv3 PMC slots':
FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ counters go in a third; never together with tracing).

    python profiles/make_counters.py OUT.json WORKLOAD_KEY STEPS "workload text" FETCH.csv WRITE.csv SQ.csv [STATS.csv]

STEPS = number of bench steps the profiled command ran (warm-up included), so that kernels launched several times
per step (pyr_resize x11, the RANSAC group) get a launches_per_step factor.  OUT.json is updated in place (one
entry per WORKLOAD_KEY); bench.py reads it for roofline.traffic and roofline.valu.insts.

Units / corrections: FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE shows half of the bytes
of wide coalesced reads (guide; re-checked in round 1 with a known 3.314 GB stream) -> x2; WRITE_SIZE is exact
for 16-B-per-lane streaming stores.  SQ_INSTS_VALU counts wave-level vector instructions.  Values are averages
over the launches of a kernel in the profiled run.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def kname(s):
    s = s.split("(")[0].replace("void ", "").strip()
    return re.sub(r"<.*$", "", s)


def per_kernel(path):
    """{kernel: {counter: (sum over dispatches, dispatch count)}}; a counter value of one dispatch may be split over rows
    (one per XCD / SE): rows are summed per Dispatch_Id first"""
    disp = defaultdict(float)
    names = {}
    for r in csv.DictReader(open(path)):
        key = (r["Dispatch_Id"], r["Counter_Name"])
        disp[key] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = kname(r["Kernel_Name"])
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for (d, c), v in disp.items():
        a = acc[names[d]][c]
        a[0] += v; a[1] += 1
    return acc


def main():
    out, key, steps, text, fetch_csv, write_csv, sq_csv = sys.argv[1:8]
    stats_csv = sys.argv[8] if len(sys.argv) > 8 else None
    steps = int(steps)
    f, w, q = per_kernel(fetch_csv), per_kernel(write_csv), per_kernel(sq_csv)
    kernels = {}
    for k in sorted(set(f) | set(w) | set(q)):
        if k.startswith("__amd") or k.startswith("calib") or k.startswith("valu_calib"):
            continue
        e = {}
        fs, fn = f.get(k, {}).get("FETCH_SIZE", (0.0, 0)); ws, wn = w.get(k, {}).get("WRITE_SIZE", (0.0, 0))
        if fn:
            e["FETCH_SIZE_KB_avg_per_launch"] = fs / fn; e["fetch_bytes_corrected"] = fs / fn * 1024 * 2
        if wn:
            e["WRITE_SIZE_KB_avg_per_launch"] = ws / wn; e["write_bytes"] = ws / wn * 1024
        if fn or wn:
            e["hbm_bytes_per_launch"] = e.get("fetch_bytes_corrected", 0.0) + e.get("write_bytes", 0.0)
        for c, (s, n) in q.get(k, {}).items():
            e[c + "_avg_per_launch"] = s / n
            e["launches_in_sq_pass"] = n
        if "SQ_INSTS_VALU_avg_per_launch" in e:
            e["valu_insts_per_launch"] = e["SQ_INSTS_VALU_avg_per_launch"]
        n_l = max(fn, wn, e.get("launches_in_sq_pass", 0))
        e["launches_per_step"] = max(1, round(n_l / steps))
        kernels[k] = e
    if stats_csv:
        tot = defaultdict(lambda: [0.0, 0])              # template instantiations of one kernel are one entry here
        for r in csv.DictReader(open(stats_csv)):
            t = tot[kname(r["Name"])]
            t[0] += float(r["TotalDurationNs"]); t[1] += int(r["Calls"])
        for k, (ns, calls) in tot.items():
            if k in kernels and calls:
                kernels[k]["kernel_trace_avg_ns"] = ns / calls; kernels[k]["kernel_trace_calls"] = calls
    db = json.load(open(out)) if os.path.exists(out) else {}
    db[key] = {"workload": text,
               "passes": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_* (three separate runs, --pmc only) "
                         "-- python3 bench.py ... --no-cpu-baseline --no-extra --no-calibrate",
               "correction": "hbm_bytes_per_launch = FETCH_SIZE KiB x 1024 x 2 (gfx950: 128-B requests tallied at 64 B) + WRITE_SIZE KiB x 1024; "
                             "counts L2 misses incl. Infinity-Cache hits",
               "kernels": kernels}
    json.dump(db, open(out, "w"), indent=1)
    tot = sum(v.get("hbm_bytes_per_launch", 0.0) * v["launches_per_step"] for v in kernels.values())
    ins = sum(v.get("valu_insts_per_launch", 0.0) * v["launches_per_step"] for v in kernels.values())
    print(f"wrote {out}[{key}]: {len(kernels)} kernels, HBM traffic per step {tot / 1e9:.2f} GB, VALU wave-instructions per step {ins / 1e9:.2f} G")


if __name__ == "__main__":
    main()
