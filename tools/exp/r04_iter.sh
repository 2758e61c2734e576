#!/bin/bash
# one measurement round of the sliced schedule: kernel traces (config 4 share, 40 chunks), then timings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sliced_schedule or pipelined_batch" > gpurun_out/r04f/t1.log 2>&1 || { tail -n 20 gpurun_out/r04f/t1.log; exit 1; }
for w in "4:tools/exp/cfg4_share.py 16" "40:tools/chunked_dickens.py 40"; do
  tag=${w%%:*}; cmd=${w#*:}
  rm -rf gpurun_out/kt; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 $cmd > gpurun_out/r04f/prof$tag.log 2>&1
  for f in gpurun_out/kt/*/*kernel_stats.csv; do [ -f "$f" ] && cp "$f" gpurun_out/r04f/k${tag}_kernel_stats.csv; done; rm -rf gpurun_out/kt
done
grep "x 8 MiB\|chunks x" gpurun_out/r04f/prof4.log gpurun_out/r04f/prof40.log | cut -c1-230
python3 - <<'P'
import csv
for name in ("k4", "k40"):
    rows = list(csv.DictReader(open(f"gpurun_out/r04f/{name}_kernel_stats.csv")))
    print(name)
    for r in rows[:16]:
        print(f"  {r['Name'][:64]:64s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_us {float(r['AverageNs'])/1e3:9.1f}")
P
for m in "0.5" "0.25,0.5,0.75"; do export X3H_SLICE_MARKS=$m; echo "marks $m"; timeout -k 10 200 python tools/chunked_dickens.py 16 40 64 96 2>&1 | grep chunks | cut -c1-200; done
