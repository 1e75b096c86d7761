cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_multirank.py -x -q -m gpu 2>&1 | tail -2
for d in 1 0 1 0; do
PANGULU_HIP_DENSIFY_AHEAD=$d timeout 600 python bench.py --no-cpu-baseline --steps 8 --warmup 3 --no-profile-pass 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ahead=$d', 'ms %.1f' % d['ms_per_step'], 'GF/s %.0f' % d['value'], 'res %.1e' % d['residual'])"
done
