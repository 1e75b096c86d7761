cd $GRAFT_REPO_ROOT
for t in 256 128 64 256 64; do
echo "DENSIFY_THREADS=$t"
PANGULU_HIP_DENSIFY_THREADS=$t SWEEP_REPS=10 timeout 300 python tools/sweep_env.py PANGULU_AMD_PANEL_LOOKAHEAD 1 2>&1 | grep -v amdgpu.ids | sed 's/ms \[.*\] min/min/'
done
