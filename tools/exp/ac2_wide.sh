#!/bin/bash
# experiment: the coder chain launched as one wavefront per workgroup against four wavefronts (= streams) per workgroup
# usage: bash tools/exp/ac2_wide.sh   (writes gpurun_out/ac2_wide.txt)
mkdir -p gpurun_out
for w in 0 1; do
  echo "== X3H_AC2_WIDE=$w" >> gpurun_out/ac2_wide.txt
  X3H_AC2_WIDE=$w python tools/exp/coder_contention.py >> gpurun_out/ac2_wide.txt 2>&1 || exit 1
  X3H_AC2_WIDE=$w python tools/many_chunks_check.py 256 256 mix 2>&1 | grep "run 2" >> gpurun_out/ac2_wide.txt
done
