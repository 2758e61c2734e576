#!/bin/bash
# N fresh `x3 -z` processes on a small file, then N fresh `x3 -d`: every exit status that is not 0 is reported with the process's stderr
# (the CLI prints the phase and the stack of a fatal signal).  usage: cli_exit_loop.sh N [tag]    env X3_CLI_RUNTIME_TEARDOWN=1: leave through exit()
N=${1:-1000}; TAG=${2:-loop}
X3=$GRAFT_REPO_ROOT/x3_compressor_amd/csrc/x3
D=$(mktemp -d); OUT=$GRAFT_REPO_ROOT/gpurun_out/cli_exit_${TAG}.txt
python3 -c "import sys; sys.path.insert(0,'$GRAFT_REPO_ROOT'); from x3_compressor_amd import synth; open('$D/in','wb').write(synth.english_like(65536).tobytes())"
bad=0; t0=$(date +%s.%N)
for i in $(seq 1 $N); do
  $X3 -z -f -w 8 -t 16 $D/in $D/out.x3 2> $D/err; rc=$?
  if [ $rc -ne 0 ]; then bad=$((bad+1)); echo "run $i: x3 -z exit status $rc" >> $OUT; cat $D/err >> $OUT; fi
  $X3 -d -f $D/out.x3 $D/back 2> $D/err; rc=$?
  if [ $rc -ne 0 ]; then bad=$((bad+1)); echo "run $i: x3 -d exit status $rc" >> $OUT; cat $D/err >> $OUT; fi
  if [ $((i % 100)) -eq 0 ]; then echo "$i runs, $bad bad" ; fi
done
cmp $D/in $D/back && echo "last round trip identical" >> $OUT
t1=$(date +%s.%N)
echo "$N x (x3 -z + x3 -d) fresh processes (X3_CLI_RUNTIME_TEARDOWN=${X3_CLI_RUNTIME_TEARDOWN:-unset}): $bad with a non-zero exit status, $(echo "$t1 - $t0" | bc) s" | tee -a $OUT
rm -rf $D
