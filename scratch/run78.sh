cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 PANGULU_AMD_STALL_S=30
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for d in 1 0 1; do
PANGULU_HIP_GETRF_LOOKAHEAD=$d timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('lookahead=$d ms %.1f' % d['ms_per_step'], 'GF/s %.0f' % d['value'], {k: v['ms'] for k, v in d['kernels'].items() if k == 'getrf'}, 'res %.3e' % d['residual'])"
done
PANGULU_HIP_DEBUG_GETRF=1 timeout 200 python tools/sweep_opt.py 2 10 2>&1 | grep "getrf stamps" | tail -1
