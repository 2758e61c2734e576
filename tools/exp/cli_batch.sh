# x3 -z --chunk-kib 256 on a 256 MiB text file by sub-batch size (--batch-mib): process wall time of consecutive fresh processes
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
for mib in 64 128 256 64; do for i in 1 2 3; do
  s=$(date +%s%N); X3H_DEBUG=1 $X3 -z -f -w 64 -t 256 --chunk-kib 256 --batch-mib $mib $D/in.bin $D/out.x3c 2> $D/err; e=$(date +%s%N)
  echo "--batch-mib $mib process $i: wall $(( (e - s) / 1000000 )) ms | $(grep -E '\[x3] ms' $D/err | tr '\n' ' ')"
done; done
s=$(date +%s%N); $X3 -d -f $D/out.x3c $D/back.bin; e=$(date +%s%N); echo "x3 -d: wall $(( (e - s) / 1000000 )) ms"; cmp $D/in.bin $D/back.bin && echo "round trip ok"
rm -rf $D
