#!/bin/bash
# mid-size batches: stage B of a slice on the parse stream (X3H_SLICE_BSTREAM=1) with three and four slices, against the default
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04m
for cfg in "0 default" "1 default" "1 0.2,0.6" "1 0.15,0.45,0.75" "1 0.1,0.3,0.6" "1 0.3,0.65"; do
  set -- $cfg
  export X3H_SLICE_BSTREAM=$1
  if [ "$2" = default ]; then unset X3H_SLICE_MARKS; else export X3H_SLICE_MARKS=$2; fi
  echo "== bstream $1 marks $2"
  timeout -k 10 120 python3 tools/chunked_dickens.py 24 32 40 48 64 96 2>/dev/null | awk '{print $1, $2, $6, $7, $9, $10, $12, $13}'
done | tee gpurun_out/r04m/bstream2.txt
