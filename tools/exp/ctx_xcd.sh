#!/bin/bash
# experiment: wavefronts per stream in x3_ctxseg_kernel (X3H_CTX_SUB) with and without keeping a stream on one XCD (X3H_CTX_XCD)
# usage: bash tools/exp/ctx_xcd.sh [text|mix] [total MiB] [chunk KiB] "subs" "xcds"  (appends to gpurun_out/ctx_xcd.txt)
mkdir -p gpurun_out
for x in ${5:-0 1}; do for n in ${4:-4 16 64 256}; do
  echo "== ${1:-text} ${2:-256} MiB in ${3:-256} KiB chunks X3H_CTX_XCD=$x X3H_CTX_SUB=$n" >> gpurun_out/ctx_xcd.txt
  X3H_CTX_XCD=$x X3H_CTX_SUB=$n python tools/many_chunks_check.py ${2:-256} ${3:-256} ${1:-text} 2>&1 | grep "run 2" >> gpurun_out/ctx_xcd.txt
done; done
