#!/bin/bash
# mid-size batches: the coder on the top 32 / 64 CUs of the mask, everything else on the rest (experiment build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04m
for m in "none none default" "top32 rest224 default" "top64 rest192 default" "top32 rest224 0.2,0.6" "top64 rest192 0.2,0.6" "top64 rest192 0.15,0.45,0.75"; do
  set -- $m
  if [ "$1" = none ]; then unset X3H_CODER_CUS; else export X3H_CODER_CUS=$1; fi
  if [ "$2" = none ]; then unset X3H_FEATURE_CUS; else export X3H_FEATURE_CUS=$2; fi
  if [ "$3" = default ]; then unset X3H_SLICE_MARKS; else export X3H_SLICE_MARKS=$3; fi
  echo "== coder CUs: $1, feature + parse CUs: $2, marks $3"
  timeout -k 10 150 python3 tools/chunked_dickens.py 1 16 24 32 40 48 64 96 2>/dev/null | awk '{print $1, $2, $6, $7, $9, $10, $12, $13}'
done | tee gpurun_out/r04m/cumask4.txt
