#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "per_stream_sort or batch_above or many_streams or stream_kernels" 2>&1 | tail -2 || exit 1
for v in 0 1; do echo "groups made by the sort: $v"; X3H_SEGSORT_GEN=$v python3 tools/many_chunks_check.py 256 256 mix 2>&1 | grep -a "MB/s" | tail -1; done
