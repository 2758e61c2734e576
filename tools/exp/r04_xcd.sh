#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "per_stream_sort or batch_above or many_streams" 2>&1 | tail -2 || exit 1
python3 tools/many_chunks_check.py 256 256 mix 2>&1 | grep -a "MB/s" | tail -1
rm -rf gpurun_out/km; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/km -- python3 tools/many_chunks_check.py 256 256 mix > gpurun_out/km_run.txt 2> gpurun_out/km.err
python3 - <<'P'
import csv, glob
f = glob.glob("gpurun_out/km/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "foreach" in n and float(r["Percentage"]) > 0.3: print(n.split("lambda")[-1][:40], r["Calls"], round(float(r["AverageNs"]) / 1e6, 3))
P
rm -rf gpurun_out/km
