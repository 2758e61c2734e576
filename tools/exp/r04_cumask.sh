#!/bin/bash
# mid-size batches (and config 4's share): the coder stream restricted to a part of the CUs (X3H_CODER_CUS)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04m
for m in none hi lo odd even q3; do
  if [ "$m" = none ]; then unset X3H_CODER_CUS; else export X3H_CODER_CUS=$m; fi
  echo "== coder CUs: $m"
  timeout -k 10 150 python3 tools/chunked_dickens.py 1 16 32 40 48 64 96 2>/dev/null | awk '{print $1, $2, $6, $7, $9, $10, $12, $13}'
done | tee gpurun_out/r04m/cumask.txt
