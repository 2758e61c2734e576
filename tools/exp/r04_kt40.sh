#!/bin/bash
# kernel trace of the 40-chunk batch (K3 in slices): which kernels of which slice take the time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04m
rm -rf gpurun_out/kt40; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt40 -- python3 tools/chunked_dickens.py 40 > gpurun_out/r04m/kt40.log 2>&1
F=$(find gpurun_out/kt40 -name '*kernel_trace.csv' | head -1)
python3 - "$F" <<'PY' | tee gpurun_out/r04m/kt40_last_call.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call = from the last x3_segscan / scan kernel on
starts = [i for i, r in enumerate(rows) if "stage_inputs" in r["Kernel_Name"]]
rows = rows[starts[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-46:]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if e - s >= 0.03: print(f"{s:8.3f} {e:8.3f} {e - s:7.3f}  {n}")
PY
rm -rf gpurun_out/kt40
