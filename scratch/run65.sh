cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.log; python -c "
import json; d=json.loads(open('gpurun_out/bench_default.log').read()); print('N=1 ms %.1f GF/s %.0f' % (d['ms_per_step'], d['value']), d['roofline']['frac'], d['cpu_baseline']['value'])"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 2>&1 | grep metric | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('N=2 (shared GPU) ms %.1f' % d['ms_per_step'], d['config']['parallelism'], d['config']['transport'], 'res %.1e' % d['residual'])"
