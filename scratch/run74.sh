cd $GRAFT_REPO_ROOT
for f in 2.0 0.95 0.8 2.0 0.95; do
PANGULU_HIP_TRSM_STAGED_FROM=$f timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('staged_from=$f', 'ms %.1f' % d['ms_per_step'], 'GF/s %.0f' % d['value'], {k: (v['ms'], v['launches']) for k, v in d['kernels'].items() if k == 'tstrf'})"
done
