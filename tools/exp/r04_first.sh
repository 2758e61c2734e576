#!/bin/bash
# round 4, first GPU call: the new tests, then the chunk-count sweep under the schedules that exist
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04a
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "size_estimates or rccl" > gpurun_out/r04a/t1.log 2>&1 && \
python -m pytest tests/test_gpu_cli.py -m gpu -x -q -k "statistics or container_the_reference" > gpurun_out/r04a/t2.log 2>&1 && \
python -m pytest tests/test_gpu_golden_sha.py -m gpu -x -q -k "config4_share or rccl_over" > gpurun_out/r04a/t3.log 2>&1
echo "tests rc $?" > gpurun_out/r04a/rc.txt
python tools/chunked_dickens.py 16 24 32 40 48 56 64 96 128 > gpurun_out/r04a/sweep_default.txt 2>&1
X3H_PIPE_STREAMS=64 X3H_PIPE_MIN=131072 python tools/chunked_dickens.py 16 24 32 40 48 56 64 > gpurun_out/r04a/sweep_pipe64.txt 2>&1
X3H_PIPE_MIN=0 python tools/chunked_dickens.py 16 24 32 40 48 > gpurun_out/r04a/sweep_nopipe.txt 2>&1
X3H_STREAM_KERNELS=1 X3H_PIPE_MIN=0 python tools/chunked_dickens.py 16 24 32 40 48 > gpurun_out/r04a/sweep_streamk.txt 2>&1
tail -3 gpurun_out/r04a/t*.log; cat gpurun_out/r04a/rc.txt gpurun_out/r04a/sweep_*.txt
