#!/bin/bash
# the same-bytes sweep inside bench.py against tools/chunked_dickens.py in one session, with the time line of the 96-chunk batch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04m
X3H_DEBUG=1 timeout -k 10 300 python3 bench.py --no-cpu --no-config35 --many-chunks-mib 0 2> gpurun_out/r04m/bsw_bench.err | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench :', [(e['chunks'], e['value'], e['ms']) for e in l['chunked_same_bytes']['sweep']])"
grep "sliced: 4 slices" gpurun_out/r04m/bsw_bench.err | tail -12 | cut -c1-330
X3H_DEBUG=1 timeout -k 10 200 python3 tools/chunked_dickens.py 64 80 96 2> gpurun_out/r04m/bsw_script.err | cut -c1-60
grep "sliced: 4 slices" gpurun_out/r04m/bsw_script.err | tail -12 | cut -c1-330
