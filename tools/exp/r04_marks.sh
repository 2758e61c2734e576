#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04d
for m in default "0.25,0.5,0.75" "0.12,0.35,0.6,0.82" "0.06,0.16,0.3,0.47,0.65,0.83" "0.5"; do
  if [ "$m" = default ]; then unset X3H_SLICE_MARKS; else export X3H_SLICE_MARKS=$m; fi
  echo "marks $m"; timeout -k 10 200 python tools/chunked_dickens.py 16 32 40 48 64 96 2>&1 | grep chunks | cut -c1-200
done > gpurun_out/r04d/marks.txt 2>&1
unset X3H_SLICE_MARKS
X3H_DEBUG=1 timeout -k 10 200 python tools/chunked_dickens.py 1 2> gpurun_out/r04d/dbg1.txt | grep chunks > gpurun_out/r04d/one.txt
timeout -k 10 300 python tools/exp/cfg4_share.py 16 > gpurun_out/r04d/cfg4.txt 2>&1
X3H_SLICED=0 timeout -k 10 300 python tools/exp/cfg4_share.py 16 > gpurun_out/r04d/cfg4_old.txt 2>&1
cat gpurun_out/r04d/marks.txt gpurun_out/r04d/one.txt; grep "sliced:" gpurun_out/r04d/dbg1.txt | tail -n 1 | cut -c1-900; grep -v amdgpu.ids gpurun_out/r04d/cfg4.txt gpurun_out/r04d/cfg4_old.txt
