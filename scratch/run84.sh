cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 PANGULU_AMD_STALL_S=60
for n in 2 8; do
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus $n --steps 2 --warmup 1 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('N=%d (one shared GPU) ms %.1f' % (d['n_gpus'], d['ms_per_step']), d['config']['transport'], 'res %.1e' % d['residual'], d['scaling'])"
done
