#!/bin/bash
# randomised differential runs against the oracle at HEAD (compress under every schedule incl. the slices, batch and single decodes): three seeds, 150 s each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04f
for seed in ${SEEDS:-404 4141 40404}; do
  timeout -k 10 260 python3 tools/fuzz_parity.py 150 $seed > gpurun_out/r04f/r04_fuzz_$seed.txt 2>&1; echo "seed $seed rc $?"; tail -n 2 gpurun_out/r04f/r04_fuzz_$seed.txt
done
