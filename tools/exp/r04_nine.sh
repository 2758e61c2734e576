#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "per_stream_sort or batch_above or many_streams" 2>&1 | tail -2 || exit 1
for v in 0 1; do echo "one pass of nine bits: $v"; X3H_SEGSORT_NINE=$v X3H_DEBUG=1 python3 tools/many_chunks_check.py 256 256 mix 2>&1 | grep -a "MB/s\|per-stream sort" | tail -3; done
