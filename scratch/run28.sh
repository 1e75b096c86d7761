cd $GRAFT_REPO_ROOT
PANGULU_HIP_DEBUG_GETRF=1 timeout 300 python tools/sweep_opt.py 2 10 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
