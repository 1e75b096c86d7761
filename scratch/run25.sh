cd $GRAFT_REPO_ROOT
timeout 300 python tools/sweep_env.py PANGULU_AMD_PANEL_LOOKAHEAD 0 1 0 1 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
