cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_multirank.py -x -q -m gpu 2>&1 | tail -3
export PANGULU_AMD_STALL_S=20 PANGULU_AMD_TRACE=1
for n in 2 4; do
timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $n --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ipc_n$n.log 2>&1
echo "N=$n rc $?"; grep "subtree mapping" gpurun_out/ipc_n$n.log | head -2; grep "metric" gpurun_out/ipc_n$n.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['n_gpus'], d['ms_per_step'], d['batches_per_step'], d['config']['transport'], d['residual'], {k: (v['launches'], v['ms']) for k, v in d['kernels'].items()})"
done
