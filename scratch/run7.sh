cd $GRAFT_REPO_ROOT
timeout 600 python tools/sweep_opt.py 2 60 40 25 10 0 --set 9=30 --set 8=6 2>&1 | tail -5
timeout 600 python tools/sweep_opt.py 8 6 8 12 16 --set 9=30 --set 2=40 2>&1 | tail -4
timeout 600 python tools/sweep_opt.py 9 30 15 0 --set 8=8 --set 2=40 2>&1 | tail -3
