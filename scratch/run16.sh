cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for la in 8 1000; do
PANGULU_AMD_LOOKAHEAD_MAX_GETRF=$la timeout 300 python tools/sweep_opt.py 2 10 2>&1 | tail -1
done
