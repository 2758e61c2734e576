#!/usr/bin/env python3
"""Aggregate a rocprofv3 *_kernel_stats.csv into per-family ms/step (usage: agg_kernel_stats.py file.csv nsteps)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = {}
for r in rows:
    n = r["Name"]
    if "x3_foreach" in n:
        key = "foreach:" + n.split("x3_foreach_kernel<")[1].split("(")[0]
    elif "rocprim" in n:
        key = "rocprim:" + ("sort" if "radix" in n else "scan" if "scan" in n else "other")
    else:
        key = n.split("(")[0][:44]
    a = agg.setdefault(key, [0.0, 0])
    a[0] += float(r["TotalDurationNs"]); a[1] += int(r["Calls"])
for k, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0]):
    print(f"{k:46s} {t / steps / 1e6:11.3f} ms/step   {c / steps:7.1f} launches/step")
