set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 300 python tools/sweep_opt.py 11 512 0 2048 2>&1 | tail -4
PANGULU_AMD_LOOKAHEAD_MAX_GETRF=0 timeout 300 python tools/sweep_opt.py 11 512 0 2>&1 | tail -3
PANGULU_AMD_LOOKAHEAD_MAX_GETRF=8 timeout 300 python tools/sweep_opt.py 11 512 2>&1 | tail -2
PANGULU_AMD_LOOKAHEAD_MAX_GETRF=1000 timeout 300 python tools/sweep_opt.py 11 512 2>&1 | tail -2
