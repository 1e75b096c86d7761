cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 PANGULU_AMD_STALL_S=30
for i in 1 2 3; do timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -1; done
for w in "fem27 --size 48" "poisson --size 64" "shell --size 200 200 --nb 128"; do
timeout 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['workload'], 'GF/s %.0f' % d['value'], 'res %.1e' % d['residual'])"
done
for d in 1 0 1 0; do
PANGULU_HIP_GETRF_LOOKAHEAD=$d timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('lookahead=$d ms %.1f' % d['ms_per_step'], 'GF/s %.0f' % d['value'], {k: v['ms'] for k, v in d['kernels'].items() if k == 'getrf'}, 'res %.1e' % d['residual'])"
done
