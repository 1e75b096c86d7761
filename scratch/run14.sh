cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
export PANGULU_AMD_STALL_S=20
for cfg in "64 600 150" "256 2000 300" "512 5000 600" "16 200 60"; do
set -- $cfg
export PANGULU_AMD_GATHER_MIN_BATCH=$1 PANGULU_AMD_GATHER_MAX_US=$2 PANGULU_AMD_GATHER_QUIET_US=$3
echo "=== gather $cfg"
timeout 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass > gpurun_out/ipc_g.log 2>&1
echo "rc $?"; grep "metric" gpurun_out/ipc_g.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['ms_per_step'], d['batches_per_step'], d['host_sched_s_last_step'])"
done
