cd $GRAFT_REPO_ROOT
for q in 4 8 8 4; do for d in 1 0; do
echo "GPU_MAX_HW_QUEUES=$q AHEAD=$d"
GPU_MAX_HW_QUEUES=$q PANGULU_HIP_DENSIFY_AHEAD=$d SWEEP_REPS=10 timeout 300 python tools/sweep_env.py PANGULU_AMD_PANEL_LOOKAHEAD 1 2>&1 | grep -v amdgpu.ids | sed 's/ms \[.*\] min/min/'
done; done
