cd $GRAFT_REPO_ROOT
for w in "poisson --size 96" "fem27 --size 64"; do
echo "=== $w"
timeout 900 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print(d['config']['workload'], 'n', d['config']['n'], 'symbolic nnz %.3g' % d['config']['symbolic_nnz'], 'flop %.3g' % d['config']['flop'], 'GF/s %.0f' % d['value'], 'ms %.1f' % d['ms_per_step'], 'res', d['residual'], 'init_s', d['init_s'], d['roofline'])
    except Exception as e:
        print('PARSE FAIL', l[:300])"
done
