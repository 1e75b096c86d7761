cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
timeout 300 python tools/sweep_opt.py 2 10 10 2>&1 | tail -2
PANGULU_AMD_LAZY_MIRRORS=1 timeout 300 python tools/sweep_opt.py 2 10 2>&1 | tail -1
