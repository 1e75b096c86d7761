cd $GRAFT_REPO_ROOT
for d in 1 0 1 0 1 0; do
PANGULU_HIP_OCCUPANCY_SUMMARIES=$d timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('summaries=$d', 'ms %.1f' % d['ms_per_step'], 'GF/s %.0f' % d['value'], 'sched %.4f' % d['host_sched_s_last_step'], {k: (v['ms'], v['launches']) for k, v in d['kernels'].items() if 'dense' in k or k == 'tstrf'}, 'init', d['init_s'])"
done
