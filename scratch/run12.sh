cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_multirank.py -x -q -m gpu 2>&1 | tail -12
export PANGULU_AMD_STALL_S=20
for sz in 398; do
echo "=== size $sz"
timeout 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --size $sz $sz > gpurun_out/ipc_$sz.log 2>&1
echo "rc $?"; grep "metric" gpurun_out/ipc_$sz.log | tail -30 | cut -c1-2500
done
