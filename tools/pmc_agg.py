#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc passes (one counter per pass) into HBM bytes per launch for the named kernels.
usage: pmc_agg.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json  (values are KB; see MI355X_MICROARCH.md, HBM)"""
import csv, json, sys
from collections import defaultdict

NAMED = ["x3_ac2_kernel", "x3_parse_kernel", "x3_modes_kernel", "x3_walk_kernel"]


def load(path, counter):
    per, other = defaultdict(lambda: [0.0, 0]), 0.0
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        name, v = r["Kernel_Name"], float(r["Counter_Value"])
        for k in NAMED:
            if name.startswith(k):
                per[k][0] += v; per[k][1] += 1
                break
        else:
            other += v
    return per, other


fetch, fo = load(sys.argv[1], "FETCH_SIZE")
write, wo = load(sys.argv[2], "WRITE_SIZE")
steps = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0   # warm-up + timed step
out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu --chunks 0 --many-chunks-mib 0 (two separate passes)",
       "units": "FETCH_SIZE / WRITE_SIZE are KB.  hbm_bytes_per_step_raw = (FETCH + WRITE) * 1024 summed over the kernel's launches of one step; "
                "hbm_bytes_per_step = (2 * FETCH + WRITE) * 1024 applies the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts half of a wide "
                "coalesced streaming read) -- valid for the vector-memory kernels; x3_ac2_kernel reads 64-byte scalar-cache lines and touches one dword per "
                "128 bytes with its L2-warming vector loads, an access width that guide calls uncalibrated: both figures are given.",
       "kernels": {}}
for k in NAMED:
    if k in fetch or k in write:
        f, nf = fetch.get(k, [0, 0]); w, nw = write.get(k, [0, 0])
        out["kernels"][k] = {"launches_per_step": nf / steps, "FETCH_SIZE_KB_per_step": round(f / steps, 1), "WRITE_SIZE_KB_per_step": round(w / steps, 1),
                             "hbm_bytes_per_step": int((2 * f + w) * 1024 / steps), "hbm_bytes_per_step_raw": int((f + w) * 1024 / steps)}
out["all_other_kernels_per_step"] = {"FETCH_SIZE_KB": round(fo / steps, 1), "WRITE_SIZE_KB": round(wo / steps, 1)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
