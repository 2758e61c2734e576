cd $GRAFT_REPO_ROOT
D=$(mktemp -d -p /tmp)
python3 - <<PY
import sys; sys.path.insert(0,'.')
import numpy as np
from x3_compressor_amd import synth
base = synth.english_like(8 << 20)
np.tile(base, 32).tofile("$D/in.bin")
PY
X3=x3_compressor_amd/csrc/x3
for i in 1 2; do X3H_DEBUG=1 $X3 -z -f -w 64 -t 256 --chunk-kib 256 $D/in.bin $D/out.x3c 2>&1 | grep -E "\[x3h\] call|\[x3\] ms" ; done
rm -rf $D
