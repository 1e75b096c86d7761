cd $GRAFT_REPO_ROOT
for a in "--no-cpu-baseline" "--no-cpu-baseline --host-threads 32" "--cpu-sample 60 60" "--cpu-sample 60 60 --host-threads 32"; do
echo "== $a"
timeout 600 python bench.py --no-profile-pass $a 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms %.1f' % d['ms_per_step'], 'GF/s %.0f' % d['value'], 'sched', d['host_sched_s_last_step'])"
done
