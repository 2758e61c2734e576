#!/bin/bash
# decoder: single-stream times, SQ counters per parse step, the 1024-stream batch (no tests; r04_dec.sh runs those)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04q
timeout -k 10 300 python3 tools/exp/cfg5_dec.py > gpurun_out/r04q/cfg5.txt 2>&1; echo "cfg5 rc $?"; cat gpurun_out/r04q/cfg5.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/r04q/pmc3 -- python3 tools/exp/dec_only.py > gpurun_out/r04q/pmc3.txt 2>&1
python3 - <<PY
import csv, glob, collections, re
steps = int(re.search(r"steps (\d+)", open("gpurun_out/r04q/pmc3.txt").read()).group(1))
for f in sorted(glob.glob("gpurun_out/r04q/pmc3/**/*counter_collection.csv", recursive=True))[-1:]:
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "x3_decode" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print({k: round(v / steps, 1) for k, v in sorted(acc.items())})
PY
timeout -k 10 300 python3 tools/exp/many_decode.py 2>&1 | tail -3
