# A/B of the many-chunk batch on ONE box: the round-4 library (tools/ab/libx3hip_r04.so, built from commit 616653c) against the current one, alternating
for i in 1 2; do
  for lib in tools/ab/libx3hip_r04.so x3_compressor_amd/csrc/libx3hip.so; do
    echo "== $lib"; X3HIP_LIBRARY=$lib MC_RUNS=4 python3 tools/many_chunks_check.py 256 256 mix 2>/dev/null | tail -2
  done
done
