#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "per_chunk_scan or dense or mr_like" 2>&1 | tail -2 || exit 1
python3 tools/many_chunks_check.py 64 256 mr 2>&1 | grep -a "MB/s" | tail -1
