#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for D in ${DIAGS:-0}; do
rm -rf gpurun_out/km; X3H_SEGSORT_DIAG=$D rocprofv3 --kernel-trace --output-format csv -d gpurun_out/km -- python3 tools/many_chunks_check.py 256 256 ${KIND:-mix} > gpurun_out/km_run.txt 2> gpurun_out/km.err
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/km/**/*kernel_trace.csv", recursive=True)[0]
print([round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 2) for r in csv.DictReader(open(f)) if "segsort" in r["Kernel_Name"]])
PY
done
rm -rf gpurun_out/km
