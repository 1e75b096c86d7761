cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 PANGULU_AMD_STALL_S=20 PANGULU_AMD_TRACE=1 PANGULU_TEST_RANK_TIMEOUT=60
for i in $(seq 1 25); do
timeout 150 python -m pytest tests/test_multirank.py -x -q -m gpu -k "peer_copies and shell_8x7" 2>&1 | tail -60 > gpurun_out/mr_dbg_$i.log
if grep -q "1 passed" gpurun_out/mr_dbg_$i.log; then rm gpurun_out/mr_dbg_$i.log; echo "ok $i"; else echo "FAIL $i"; fi
done
