cd $GRAFT_REPO_ROOT
for nf in 1000000 129 1; do
echo "NARROW_FROM=$nf"
PANGULU_HIP_GETRF_NARROW_FROM=$nf timeout 300 python tools/sweep_env.py PANGULU_AMD_PANEL_LOOKAHEAD 1 0 2>&1 | grep -v amdgpu.ids
done
PANGULU_HIP_GETRF_NARROW_FROM=129 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
