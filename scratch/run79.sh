cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 PANGULU_AMD_STALL_S=25
for i in 1 2 3 4 5 6 7 8 9 10; do
timeout 200 python -m pytest tests/test_multirank.py -x -q -m gpu 2>&1 | tail -1
done
for i in 1 2 3; do timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1; done
