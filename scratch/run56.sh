cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 PANGULU_AMD_STALL_S=25 PANGULU_HIP_TRSM_DIRECT=1 PANGULU_AMD_TRACE=1
for i in 1 2 3 4 5 6 7 8; do
timeout 600 python -m pytest tests/test_multirank.py -x -q -m gpu 2>&1 | tail -120 > gpurun_out/mr_loop_$i.log
tail -1 gpurun_out/mr_loop_$i.log
done
