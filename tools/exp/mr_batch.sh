cd $GRAFT_REPO_ROOT
python tools/many_chunks_check.py 64 256 mr 2>&1 | grep "run 2"
X3H_SEG_REFINE=0 python tools/many_chunks_check.py 64 256 mr 2>&1 | grep "run 2"
