cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 PANGULU_AMD_STALL_S=30 PANGULU_HIP_MIRROR_FRACTION=0.2
for n in 4 3; do
timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $n --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass > gpurun_out/ipc_n$n.log 2>&1
echo "N=$n rc $?"; grep "metric" gpurun_out/ipc_n$n.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['n_gpus'], d['ms_per_step'], d['batches_per_step'], d['config']['parallelism'], d['config']['transport'], d['residual'])"
grep -i "error\|fatal\|PanguLU-AMD" gpurun_out/ipc_n$n.log | head -5
done
