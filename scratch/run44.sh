cd $GRAFT_REPO_ROOT
for k in a b a b; do
cp scratch/lib_$k.so pangulu_amd/lib/libpangulu_amd_r64.so
echo "variant $k"
SWEEP_REPS=10 timeout 300 python tools/sweep_env.py PANGULU_AMD_PANEL_LOOKAHEAD 1 2>&1 | grep -v amdgpu.ids | sed 's/ms \[.*\] min/min/'
done
