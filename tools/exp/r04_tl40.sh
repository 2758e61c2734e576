#!/bin/bash
# time line of the sliced schedule on 40 chunks of the dickens-sized bytes (X3H_DEBUG=1), default marks and 0.2,0.6
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04m
for marks in default "0.2,0.6"; do
  if [ "$marks" = default ]; then unset X3H_SLICE_MARKS; else export X3H_SLICE_MARKS=$marks; fi
  echo "== marks $marks"
  X3H_DEBUG=1 timeout -k 10 120 python3 tools/chunked_dickens.py 40 2>&1 | grep -E "sliced:|chunks x" | tail -3 | cut -c1-900
done | tee gpurun_out/r04m/tl40.txt
