cd $GRAFT_REPO_ROOT
for wt in "8 15" "64 16" "64 1024" "16 64"; do set -- $wt
for d in 1 64 256 2048; do
  echo "-w $1 -t $2 X3H_WALK_DENSE=$d mix: $(MC_W=$1 MC_T=$2 X3H_WALK_DENSE=$d python tools/many_chunks_check.py 256 256 mix 2>&1 | grep 'run 2' | sed 's/.*device ms: //')"
done; done
