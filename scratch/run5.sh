cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
timeout 300 python tools/sweep_opt.py 2 25 0 2>&1 | tail -2
timeout 300 python tools/sweep_opt.py 9 30 0 2>&1 | tail -2
