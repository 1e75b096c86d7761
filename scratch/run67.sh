cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 PANGULU_AMD_STALL_S=40
for n in 3 4; do
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $n --steps 2 --warmup 1 2>&1 | grep metric | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('N=%d (shared GPU) ms %.1f' % (d['n_gpus'], d['ms_per_step']), d['config']['transport'], 'res %.1e' % d['residual'], 'cpu_baseline', d['cpu_baseline'])"
done
for t in host rccl; do
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --transport $t 2>&1 | grep "metric\|falling back" | cut -c1-200 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l + '' if l.rstrip().endswith('}') else l); print('transport asked $t ->', d['config']['transport'], 'ms %.1f' % d['ms_per_step'], 'res %.1e' % d['residual'])
    else: print(l.strip())" 2>&1 | tail -2
done
