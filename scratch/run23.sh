cd $GRAFT_REPO_ROOT
nproc; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" 
PANGULU_AMD_LOOKAHEAD_MAX_GETRF=128 PANGULU_HIP_HOST_TIMING=1 timeout 300 python tools/sweep_opt.py 2 10 2>&1 | grep -v amdgpu.ids | tail -12
