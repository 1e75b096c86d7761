cd $GRAFT_REPO_ROOT
for t in 64 0 256 1024 64; do
echo "KSPLIT_WGS=$t"
PANGULU_HIP_KSPLIT_WGS=$t SWEEP_REPS=10 timeout 300 python tools/sweep_env.py PANGULU_AMD_PANEL_LOOKAHEAD 1 2>&1 | grep -v amdgpu.ids | sed 's/ms \[.*\] min/min/'
done
