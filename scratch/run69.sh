cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_multirank.py -x -q -m gpu 2>&1 | tail -2
for d in 1 0 1 0; do
echo "SUMMARIES=$d"
PANGULU_HIP_OCCUPANCY_SUMMARIES=$d SWEEP_REPS=10 timeout 300 python tools/sweep_env.py PANGULU_AMD_PANEL_LOOKAHEAD 1 2>&1 | grep -v amdgpu.ids | sed 's/ms \[.*\] min/min/'
done
