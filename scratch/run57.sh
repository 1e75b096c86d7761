cd $GRAFT_REPO_ROOT
for d in 1 0 1 0 1 0; do
echo "TRSM_DIRECT=$d"
PANGULU_HIP_TRSM_DIRECT=$d SWEEP_REPS=10 timeout 300 python tools/sweep_env.py PANGULU_AMD_PANEL_LOOKAHEAD 1 2>&1 | grep -v amdgpu.ids | sed 's/ms \[.*\] min/min/'
done
