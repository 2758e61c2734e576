#!/bin/bash
# one measurement round of the sliced schedule: tests, kernel traces (config 4 share, 40 chunks), SQ counters of the config 4 share, timings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sliced_schedule or pipelined_batch or stream_kernels or golden" > gpurun_out/r04f/t1.log 2>&1 || { tail -n 20 gpurun_out/r04f/t1.log; exit 1; }
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
    for r in rows[:14]:
        print(f"  {r['Name'][:64]:64s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_us {float(r['AverageNs'])/1e3:9.1f}")
P
rm -rf gpurun_out/pm; timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pm -- python3 tools/exp/cfg4_share.py 16 > gpurun_out/r04f/pmc4.log 2>&1
python3 - <<'P'
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/pm/**/*counter_collection.csv", recursive=True))
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(f[-1])):
        k = r["Kernel_Name"].split("(")[0][:28]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        if k.startswith(("x3s_", "x3_ac2", "x3_emit")):
            w = v.get("SQ_WAVES", 1) or 1
            print(f"  {k:28s} waves {w:9.0f} | per wave: cycles {4*v['SQ_WAVE_CYCLES']/w:10.0f} wait_any {4*v['SQ_WAIT_ANY']/w:10.0f} wait_inst {4*v['SQ_WAIT_INST_ANY']/w:9.0f} active {4*v['SQ_ACTIVE_INST_ANY']/w:9.0f} | valu {v['SQ_INSTS_VALU']/w:8.0f} salu {v['SQ_INSTS_SALU']/w:8.0f} lds {v['SQ_INSTS_LDS']/w:7.0f}")
P
rm -rf gpurun_out/pm
for m in "0.5" "0.25,0.5,0.75"; do export X3H_SLICE_MARKS=$m; echo "marks $m"; timeout -k 10 200 python tools/chunked_dickens.py 16 40 64 96 2>&1 | grep chunks | cut -c1-200; done
