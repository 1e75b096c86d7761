cd $GRAFT_REPO_ROOT
PANGULU_HIP_GETRF_NARROW_FROM=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
for nf in 1 1000000; do
echo "NARROW_FROM=$nf"
PANGULU_HIP_DEBUG_GETRF=1 PANGULU_HIP_GETRF_NARROW_FROM=$nf SWEEP_REPS=8 timeout 300 python tools/sweep_env.py PANGULU_AMD_PANEL_LOOKAHEAD 1 2>&1 | grep -v amdgpu.ids | sed 's/ms \[.*\] min/min/'
PANGULU_HIP_DEBUG_GETRF=1 PANGULU_HIP_GETRF_NARROW_FROM=$nf timeout 300 python tools/sweep_opt.py 2 10 2>&1 | grep "getrf stamps"
done
