# A PROFILE build of the decoder for tools/dec_prof.py: decode.hip compiled with -DX3_DEC_PROFILE (section clocks -- s_memtime -- around parts of the chain's step, reported as the
# three "kcycles" figures of the X3H_DEBUG line of a decode call) and linked with the product's other objects into tools/ab/libx3hip_prof.so (git-ignored; travels with gpurun).
# Which sections are timed is decided in decode.hip / decode_hit.inc under #ifdef X3_DEC_PROFILE: move the three accumulators (pc_wait, pc_flight, pc_chain) to the sections in
# question -- that is how round 5 found the recency-list sweeps (746 cycles of a step) and the block moves of short context0 lists.  Every clock read costs the step ~50 cycles.
set -e
cd "$(dirname "$0")/../x3_compressor_amd/csrc"
make -s libx3hip.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable -DX3_DEC_PROFILE=1 -c decode.hip -o /tmp/decode_prof.o
mkdir -p ../../tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/ab/libx3hip_prof.so scan.o scan2.o scan3.o parse.o code2.o code3.o code4.o /tmp/decode_prof.o prims.o api.o x3_container.o -lpthread -ldl
echo "built tools/ab/libx3hip_prof.so -- run: python tools/dec_prof.py tools/ab/libx3hip_prof.so"
