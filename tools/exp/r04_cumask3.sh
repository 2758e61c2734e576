#!/bin/bash
# time line of 40 chunks with the coder stream and the feature / parse streams on disjoint parts of the CU mask
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04m
for m in "none none" "hi lo" "lo hi" "odd even" "even odd"; do
  set -- $m
  if [ "$1" = none ]; then unset X3H_CODER_CUS; else export X3H_CODER_CUS=$1; fi
  if [ "$2" = none ]; then unset X3H_FEATURE_CUS; else export X3H_FEATURE_CUS=$2; fi
  echo "== coder CUs: $1, feature + parse CUs: $2"
  X3H_DEBUG=1 timeout -k 10 120 python3 tools/chunked_dickens.py 40 2>&1 | grep -E "sliced:" | tail -1 | cut -c1-400
done | tee gpurun_out/r04m/cumask3.txt
