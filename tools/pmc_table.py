#!/usr/bin/env python3
"""per-kernel sums of a rocprofv3 --pmc counter_collection.csv: python tools/pmc_table.py file.csv"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); t = collections.defaultdict(float); n = collections.Counter(); seen = set()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-44:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); n[k] += 1; t[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
cn = sorted({c for k in agg for c in agg[k]})
print("kernel".ljust(46), "n".rjust(4), "ms".rjust(8), *[c.replace("SQ_", "")[:14].rjust(15) for c in cn])
for k in sorted(agg, key=lambda k: -t[k])[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(k.ljust(46), str(n[k]).rjust(4), f"{t[k]:8.2f}", *[f"{agg[k].get(c, 0):15.4g}" for c in cn])
