cd $GRAFT_REPO_ROOT
for d in 1 0 1 0; do
PANGULU_HIP_TRSM_DIRECT=$d timeout 600 python bench.py --no-cpu-baseline --steps 8 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('direct=$d', 'ms %.1f' % d['ms_per_step'], 'GF/s %.0f' % d['value'], {k: v['ms'] for k, v in d['kernels'].items()}, 'res %.1e' % d['residual'])"
done
