#!/bin/bash
# experiment: move-to-front ranks by one wavefront per stream (X3H_MTF_PAR=0) against eight time ranges per stream (default), by batch shape
mkdir -p gpurun_out
for shape in "256 1024 text" "256 512 text" "256 256 mix" "256 64 mix" "64 16 text"; do for p in 0 1; do
  echo "== $shape X3H_MTF_PAR=$p" >> gpurun_out/mtf_par.txt
  X3H_MTF_PAR=$p python tools/many_chunks_check.py $shape 2>&1 | grep "run 2" >> gpurun_out/mtf_par.txt
done; done
