cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 PANGULU_AMD_STALL_S=30
run() {
n=$1; shift
env "$@" timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $n --steps 3 --warmup 1 --no-cpu-baseline --no-profile-pass > gpurun_out/mr.log 2>&1
echo "N=$n $@ rc $?"; grep "metric" gpurun_out/mr.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   ', d['n_gpus'], 'ms %.1f' % d['ms_per_step'], 'batches', d['batches_per_step'], d['config']['transport'], 'res %.2e' % d['residual'])"
grep -i "error\|fatal" gpurun_out/mr.log | head -3
}
run 2 X=1
run 2 PANGULU_AMD_GATHER_QUIET_US=300 PANGULU_AMD_GATHER_MAX_US=2000
run 4 X=1
run 4 PANGULU_AMD_GATHER_QUIET_US=300 PANGULU_AMD_GATHER_MAX_US=2000
run 8 X=1
