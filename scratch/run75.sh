cd $GRAFT_REPO_ROOT
export PANGULU_HIP_GETRF_LOOKAHEAD=1
timeout 120 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
echo "rc $?"
timeout 120 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('lookahead ms %.1f' % d['ms_per_step'], 'GF/s %.0f' % d['value'], {k: (v['ms'], v['launches']) for k, v in d['kernels'].items() if k == 'getrf'}, 'res %.1e' % d['residual'])"
export PANGULU_HIP_GETRF_LOOKAHEAD=0
timeout 120 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('baseline  ms %.1f' % d['ms_per_step'], 'GF/s %.0f' % d['value'], {k: (v['ms'], v['launches']) for k, v in d['kernels'].items() if k == 'getrf'}, 'res %.1e' % d['residual'])"
