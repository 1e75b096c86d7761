cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --steps 8 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms %.1f' % d['ms_per_step'], 'GF/s %.0f' % d['value'], {k: v['ms'] for k, v in d['kernels'].items()}, 'res %.1e' % d['residual'])"
done
PANGULU_HIP_DEBUG_SSSSM=1 timeout 300 python tools/sweep_opt.py 2 10 2>&1 | grep "stamps" | cut -c1-420
