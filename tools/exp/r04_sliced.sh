#!/bin/bash
# round 4: K3 in slices on the GPU -- golden streams forced through it, then the workloads it is meant for
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04b
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sliced_schedule or size_estimates or pipelined_batch" > gpurun_out/r04b/t1.log 2>&1
echo "t1 rc $?" > gpurun_out/r04b/rc.txt
[ "$(cat gpurun_out/r04b/rc.txt)" = "t1 rc 0" ] || { tail -n 30 gpurun_out/r04b/t1.log; exit 1; }
X3H_DEBUG=1 timeout -k 10 300 python tools/chunked_dickens.py 1 16 40 2> gpurun_out/r04b/dbg_sliced.txt > gpurun_out/r04b/sweep_sliced_few.txt
timeout -k 10 300 python tools/chunked_dickens.py 1 16 24 32 40 48 56 64 96 128 > gpurun_out/r04b/sweep_sliced.txt 2>&1
X3H_SLICED=0 timeout -k 10 300 python tools/chunked_dickens.py 1 40 64 > gpurun_out/r04b/sweep_unsliced.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_golden_sha.py -m gpu -x -q -k "config4_share or cfg2_full or cfg4_zipf" > gpurun_out/r04b/t2.log 2>&1
echo "t2 rc $?" >> gpurun_out/r04b/rc.txt
cat gpurun_out/r04b/rc.txt; tail -n 4 gpurun_out/r04b/t1.log gpurun_out/r04b/t2.log; grep "sliced:" gpurun_out/r04b/dbg_sliced.txt | cut -c1-1500; cat gpurun_out/r04b/sweep_sliced.txt gpurun_out/r04b/sweep_unsliced.txt
