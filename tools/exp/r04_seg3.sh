#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "per_chunk_scan or many_chunk or batch_above" 2>&1 | tail -2 || exit 1
X3H_SEG_PROF=1 python3 tools/many_chunks_check.py 256 256 mix 2>&1 | grep -a "scan3\|MB/s" | tail -2
