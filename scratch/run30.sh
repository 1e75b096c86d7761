cd $GRAFT_REPO_ROOT
export SWEEP_REPS=10
for r in 1 0 1 0; do
echo "RECORDS_STREAM=$r"
PANGULU_HIP_RECORDS_STREAM=$r timeout 300 python tools/sweep_env.py PANGULU_AMD_PANEL_LOOKAHEAD 1 2>&1 | grep -v amdgpu.ids | sed 's/ms \[.*\] min/min/'
done
