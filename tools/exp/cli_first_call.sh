# experiment: HIP API time of ONE x3 -z --chunk-kib K call (first-call costs: allocations, code loading).  usage: bash tools/exp/cli_first_call.sh K
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 - <<PY
import sys
sys.path.insert(0, '.')
import numpy as np
from x3_compressor_amd import synth
base = synth.english_like(8 << 20)
np.tile(base, 32).tofile("/tmp/in256.bin")
PY
K=${1:-64}
rm -rf gpurun_out/clitrace; rm -f /tmp/c.x3c
rocprofv3 --hip-trace --stats --output-format csv -d gpurun_out/clitrace -- x3_compressor_amd/csrc/x3 -z -w 64 -t 256 --chunk-kib $K /tmp/in256.bin /tmp/c.x3c 2>&1 | grep -i "elapsed\|device ms"
for f in gpurun_out/clitrace/*/*hip_api_stats.csv; do [ -f "$f" ] && head -8 "$f"; done
