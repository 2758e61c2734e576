#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "per_stream_sort or batch_above" 2>&1 | tail -2 || exit 1
python3 tools/many_chunks_check.py 256 256 mix 2>&1 | grep -a "MB/s" | tail -1
rm -rf gpurun_out/sp; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/sp -- python3 tools/many_chunks_check.py 256 256 mix > gpurun_out/sp.txt 2>&1
python3 - <<'P'
import csv, glob, collections
f = glob.glob("gpurun_out/sp/**/*counter_collection.csv", recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "segsort" in r["Kernel_Name"] or "segscan" in r["Kernel_Name"]: acc.setdefault((r["Dispatch_Id"], r["Kernel_Name"][:20]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for d, v in list(acc.items())[-3:]: print(d, {k: "%.3g" % x for k, x in v.items()})
f = glob.glob("gpurun_out/sp/**/*kernel_trace.csv", recursive=True)[0]
print([(r["Kernel_Name"][:17], round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 2)) for r in csv.DictReader(open(f)) if "segsort" in r["Kernel_Name"] or "segscan" in r["Kernel_Name"]][-3:])
P
rm -rf gpurun_out/sp
