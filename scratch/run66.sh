cd $GRAFT_REPO_ROOT
SWEEP_REPS=10 timeout 300 python tools/sweep_env.py PANGULU_AMD_ASYNC_LAUNCH 0 1 0 1 2>&1 | grep -v amdgpu.ids | sed 's/ms \[.*\] min/min/'
PANGULU_HIP_HOST_TIMING=1 timeout 300 python tools/sweep_opt.py 2 10 2>&1 | grep "host s\|host sched" | tail -3
