cd $GRAFT_REPO_ROOT
timeout 300 python tools/sweep_env.py PANGULU_AMD_LOOKAHEAD_MAX_GETRF 8 0 32 64 128 100000 2>&1 | grep -v amdgpu.ids
