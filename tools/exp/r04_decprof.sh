#!/bin/bash
# experiment: where the decoder's chain spends its time -- tools/exp/libx3hip_prof.so is the library built with -DX3_DEC_PROFILE=1 (s_memtime around the wait for
# the requested context blocks):  for f in *.hip: hipcc $(HIPFLAGS) -DX3_DEC_PROFILE=1 -c ...; hipcc -shared -o tools/exp/libx3hip_prof.so ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04p
X3H_DEBUG=1 X3_LIB=$GRAFT_REPO_ROOT/tools/exp/libx3hip_prof.so timeout -k 10 600 python3 tools/exp/cfg5_dec.py 2>&1 | grep -v "^\[x3h\] \(sliced\|pipelined\)" | tee gpurun_out/r04p/decprof.txt
