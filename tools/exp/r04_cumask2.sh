#!/bin/bash
# mid-size batches: coder stream on one half of the CU mask, the feature / parse streams on the other (X3H_CODER_CUS, X3H_FEATURE_CUS)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04m
for m in "none none" "hi lo" "lo hi" "odd even" "q3 lo"; do
  set -- $m
  if [ "$1" = none ]; then unset X3H_CODER_CUS; else export X3H_CODER_CUS=$1; fi
  if [ "$2" = none ]; then unset X3H_FEATURE_CUS; else export X3H_FEATURE_CUS=$2; fi
  echo "== coder CUs: $1, feature + parse CUs: $2"
  timeout -k 10 150 python3 tools/chunked_dickens.py 1 16 32 40 48 64 96 2>/dev/null | awk '{print $1, $2, $6, $7, $9, $10, $12, $13}'
done | tee gpurun_out/r04m/cumask2.txt
