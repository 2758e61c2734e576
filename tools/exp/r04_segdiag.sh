#!/bin/bash
# timing-only experiments on x3_segscan_kernel (results are WRONG with any bit set): 1 no small-K path, 2 no level-counter atomics, 4 no level-1 search, 8 no level tests, 16 no dense check
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 0 1 2 4 8 16 3 7; do echo "diag $d"; X3H_SEG_DIAG=$d X3H_SEG_PROF=1 python3 tools/many_chunks_check.py 256 256 mix 2>&1 | grep -a "scan3" | tail -1; done
