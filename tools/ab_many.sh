# A/B of the many-chunk batch on ONE box: the round-4 library against the current one, alternating.  The old library is not in the tree: build it first,
#   git worktree add /tmp/r04tree 616653c && make -C /tmp/r04tree/x3_compressor_amd/csrc libx3hip.so && mkdir -p tools/ab && cp /tmp/r04tree/x3_compressor_amd/csrc/libx3hip.so tools/ab/libx3hip_r04.so
# (*.so is git-ignored but travels with gpurun)
for i in 1 2; do
  for lib in tools/ab/libx3hip_r04.so x3_compressor_amd/csrc/libx3hip.so; do
    echo "== $lib"; X3HIP_LIBRARY=$lib MC_RUNS=4 python3 tools/many_chunks_check.py 256 256 mix 2>/dev/null | tail -2
  done
done
