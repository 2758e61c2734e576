cd $GRAFT_REPO_ROOT
for d in 128 256 512 1024 2048; do
  echo "X3H_WALK_DENSE=$d mr:  $(X3H_WALK_DENSE=$d python tools/many_chunks_check.py 64 256 mr 2>&1 | grep 'run 2' | sed 's/.*device ms: //')"
  echo "X3H_WALK_DENSE=$d mix: $(X3H_WALK_DENSE=$d python tools/many_chunks_check.py 256 256 mix 2>&1 | grep 'run 2' | sed 's/.*device ms: //')"
done
