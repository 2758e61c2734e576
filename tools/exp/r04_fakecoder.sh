#!/bin/bash
# experiment: 40 chunks, the coder replaced by a kernel of the same shape (X3H_FAKE_CODER=iterations,mode: 0 no memory, 1 streaming 64-byte scalar loads, 2 scalar stores, 3 both):
# are the feature kernels beside it still slow?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04m
for it in none 4,7 1,7 1,1 1,2 1,4 7,0; do
  if [ "$it" = none ]; then unset X3H_FAKE_CODER; else export X3H_FAKE_CODER=$it; fi
  echo "== X3H_FAKE_CODER=$it"
  X3H_DEBUG=1 timeout -k 10 120 python3 tools/chunked_dickens.py 40 2>&1 | grep -E "sliced:" | tail -1 | cut -c1-400
done | tee gpurun_out/r04m/fakecoder.txt
