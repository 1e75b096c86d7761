cd $GRAFT_REPO_ROOT
PANGULU_HIP_DEBUG_SSSSM=1 timeout 300 python tools/sweep_opt.py 2 10 2>&1 | grep "stamps\|option"
