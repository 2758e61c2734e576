#!/bin/bash
# A/B of one environment switch on the many-chunk batches: bash tools/exp/ab_env.sh NAME "v1 v2" [shapes...]   (appends to gpurun_out/ab_env.txt)
mkdir -p gpurun_out
name=$1; vals=$2; shift 2
[ $# -eq 0 ] && set -- "256 256 text" "256 256 mix"
for shape in "$@"; do for v in $vals; do
  echo "== $shape $name=$v" >> gpurun_out/ab_env.txt
  env $name=$v python tools/many_chunks_check.py $shape 2>&1 | grep "run 2" >> gpurun_out/ab_env.txt
done; done
