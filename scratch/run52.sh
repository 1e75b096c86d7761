cd $GRAFT_REPO_ROOT
timeout 400 python tools/sweep_opt.py 9 5 10 20 40 80 2>&1 | grep "^option"
timeout 400 python tools/sweep_opt.py 2 5 10 20 40 2>&1 | grep "^option"
timeout 400 python tools/sweep_opt.py 8 4 8 16 32 2>&1 | grep "^option"
