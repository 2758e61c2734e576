#!/bin/bash
# mid-size batches (the dickens-sized bytes as 24-96 chunks): which checkpoint marks for K3 in slices
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04m
for marks in default "0.5" "0.25,0.6" "0.2,0.6" "0.15,0.45,0.75" "0.1,0.3,0.6" "0.12,0.3,0.5,0.75" "0.08,0.2,0.4,0.65"; do
  if [ "$marks" = default ]; then unset X3H_SLICE_MARKS; else export X3H_SLICE_MARKS=$marks; fi
  echo "== marks $marks"
  timeout -k 10 120 python3 tools/chunked_dickens.py 16 24 32 40 48 64 96 2>/dev/null | awk '{print $1, $2, $6, $7, $9, $10, $12, $13}'
done | tee gpurun_out/r04m/marks2.txt
