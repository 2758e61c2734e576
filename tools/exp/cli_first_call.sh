# the CLI's first call: N consecutive fresh `x3 -z --chunk-kib 256` processes on a 256 MiB text file, wall clock per process + the library's own
# account of where the call went (X3H_DEBUG: wall of the call, hipMalloc/hipFree share)
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
for B in ${BATCHES:-64 32 128}; do
for i in 1 2 3 4 5; do
  s=$(date +%s%N)
  X3H_DEBUG=1 $X3 -z -f -w 64 -t 256 --chunk-kib 256 --batch-mib $B $D/in.bin $D/out.x3c 2> $D/err; rc=$?
  e=$(date +%s%N)
  echo "--batch-mib $B process $i: rc $rc wall $(( (e - s) / 1000000 )) ms | $(grep -E 'elapsed time|\[x3\] ms' $D/err | tr '\n' ' ') | $(grep -E '\[x3h\] call' $D/err | sed 's/.x3h. call: //' | tr '\n' ';')"
done
for i in 1 2; do
s=$(date +%s%N); X3H_DEBUG=1 $X3 -d -f --batch-mib $B $D/out.x3c $D/back.bin 2> $D/err; e=$(date +%s%N)
echo "--batch-mib $B x3 -d: wall $(( (e - s) / 1000000 )) ms | $(grep -E 'elapsed time|\[x3\] ms' $D/err | tr '\n' ' ')"; done; cmp $D/in.bin $D/back.bin && echo "cmp clean"
done
rm -rf $D
