cd $GRAFT_REPO_ROOT
for k in 3 0; do
cp scratch/lib_k$k.so pangulu_amd/lib/libpangulu_amd_r64.so
echo "PRESTAGED=$k"
SWEEP_REPS=8 timeout 300 python tools/sweep_env.py PANGULU_AMD_PANEL_LOOKAHEAD 1 2>&1 | grep -v amdgpu.ids | sed 's/ms \[.*\] min/min/'
done
