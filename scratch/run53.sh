cd $GRAFT_REPO_ROOT
PANGULU_HIP_DEBUG_TRSM=1 timeout 300 python tools/sweep_opt.py 2 10 2>&1 | grep "trsm stamps"
