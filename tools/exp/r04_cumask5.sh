#!/bin/bash
# mid-size batches with the masked stream layout (default) and without (X3H_SLICE_CUMASK=0), under several mark sets
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04m
for m in "0 default" "1 default"; do
  set -- $m
  export X3H_SLICE_CUMASK=$1
  if [ "$2" = default ]; then unset X3H_SLICE_MARKS; else export X3H_SLICE_MARKS=$2; fi
  echo "== masked layout $1, marks $2"
  timeout -k 10 150 python3 tools/chunked_dickens.py 1 4 8 12 16 24 32 40 48 64 80 96 112 128 2>/dev/null | awk '{print $1, $2, $6, $7, $9, $10, $12, $13}'
done | tee gpurun_out/r04m/cumask5.txt
