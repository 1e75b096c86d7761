cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_multirank.py -x -q -m gpu 2>&1 | tail -3
export PANGULU_AMD_STALL_S=20
timeout 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ipc_398.log 2>&1
echo "rc $?"; grep "metric" gpurun_out/ipc_398.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['ms_per_step'], d['batches_per_step'], d['host_sched_s_last_step'], d['config']['transport'], {k: (v['launches'], v['ms']) for k, v in d['kernels'].items()})"
