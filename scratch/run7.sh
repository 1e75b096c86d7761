cd $GRAFT_REPO_ROOT
timeout 600 python tools/sweep_opt.py 8 8 12 16 32 0 2>&1 | tail -5
timeout 600 python tools/sweep_opt.py 2 10 0 30 2>&1 | tail -3
timeout 600 python tools/sweep_opt.py 11 512 128 2048 2>&1 | tail -3
