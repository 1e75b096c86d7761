cd $GRAFT_REPO_ROOT
for d in 1 0 1 0 1 0; do
echo "AHEAD=$d"
PANGULU_HIP_DENSIFY_AHEAD=$d SWEEP_REPS=12 timeout 300 python tools/sweep_env.py PANGULU_AMD_PANEL_LOOKAHEAD 1 2>&1 | grep -v amdgpu.ids | sed 's/ms \[.*\] min/min/'
done
