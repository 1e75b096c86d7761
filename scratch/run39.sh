cd $GRAFT_REPO_ROOT
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.log
python -c "
import sys, json
d = json.loads(open('gpurun_out/bench_default.log').read()); print('ms %.1f' % d['ms_per_step'], 'GF/s %.0f' % d['value'], 'sched', d['host_sched_s_last_step'], d['cpu_baseline'])"
