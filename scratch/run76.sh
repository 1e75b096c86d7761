cd $GRAFT_REPO_ROOT
PANGULU_HIP_GETRF_LOOKAHEAD=1 PANGULU_HIP_DEBUG_GETRF=1 timeout 200 python tools/sweep_opt.py 2 10 2>&1 | grep "getrf stamps" | tail -1
