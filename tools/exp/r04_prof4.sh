#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
rm -rf gpurun_out/kt; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 tools/exp/cfg4_share.py 16 > gpurun_out/r04e/prof4.log 2>&1
for f in gpurun_out/kt/*/*kernel_stats.csv; do [ -f "$f" ] && cp "$f" gpurun_out/r04e/k4_kernel_stats.csv; done; rm -rf gpurun_out/kt
grep "x 8 MiB" gpurun_out/r04e/prof4.log
python3 - <<'P'
import csv
rows = list(csv.DictReader(open("gpurun_out/r04e/k4_kernel_stats.csv")))
for r in rows[:24]:
    print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_us {float(r['AverageNs'])/1e3:9.1f}")
P
