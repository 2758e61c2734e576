#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "per_stream_sort or batch_above or many_streams or decode_batch_of_many" 2>&1 | tail -3 || exit 1
for v in 0 1; do echo "per-stream sort: $v"; X3H_SEGSORT=$v python3 tools/many_chunks_check.py 256 256 mix 2>&1 | grep -a "MB/s" | tail -1; done
bash tools/exp/r04_kt_many.sh
