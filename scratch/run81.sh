cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 PANGULU_AMD_STALL_S=25 PANGULU_TEST_RANK_TIMEOUT=120
for i in $(seq 1 12); do
timeout 300 python -m pytest tests/test_multirank.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/mr_dbg_$i.log
if grep -q "7 passed" gpurun_out/mr_dbg_$i.log; then rm gpurun_out/mr_dbg_$i.log; echo "ok $i"; else echo "FAIL $i"; fi
done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 45123 bench.py --gpus 2 --steps 2 --warmup 1 --no-profile-pass 2>&1 | grep metric | cut -c1-120
