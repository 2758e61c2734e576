#!/bin/bash
# where a slice's feature time goes: kernel trace of the 40-chunk batch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
export GPU_MAX_HW_QUEUES=8
rm -rf gpurun_out/kt; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 tools/chunked_dickens.py 40 > gpurun_out/r04c/prof40.log 2>&1
for f in gpurun_out/kt/*/*kernel_stats.csv; do [ -f "$f" ] && cp "$f" gpurun_out/r04c/k40_kernel_stats.csv; done; rm -rf gpurun_out/kt
rm -rf gpurun_out/kt; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 tools/chunked_dickens.py 1 > gpurun_out/r04c/prof1.log 2>&1
for f in gpurun_out/kt/*/*kernel_stats.csv; do [ -f "$f" ] && cp "$f" gpurun_out/r04c/k1_kernel_stats.csv; done; rm -rf gpurun_out/kt
tail -n 2 gpurun_out/r04c/prof40.log gpurun_out/r04c/prof1.log
python3 - <<'P'
import csv
for name in ("k40", "k1"):
    rows = list(csv.DictReader(open(f"gpurun_out/r04c/{name}_kernel_stats.csv")))
    print(name)
    for r in rows[:28]:
        print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_us {float(r['AverageNs'])/1e3:9.1f}")
P
