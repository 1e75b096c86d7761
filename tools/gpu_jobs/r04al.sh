#!/bin/bash
# round 4, final tree: the profile recipe on the ldoor-class matrix (restores its key in profiles/hbm_traffic.json) and its own bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
cp profiles/hbm_traffic.json gpurun_out/hbm_traffic.json
bash tools/profile_recipe.sh r04al_shell398 --workload shell > gpurun_out/r04al_profile_recipe.log 2>&1
tail -3 gpurun_out/r04al_profile_recipe.log | cut -c1-200
cp gpurun_out/hbm_traffic.json profiles/hbm_traffic.json
( time timeout 900 python bench.py --workload shell --steps 20 --warmup 5 ) > gpurun_out/r04al_bench_shell398.log 2> gpurun_out/r04al_bench_shell398.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04al_bench_shell398.log').readline())
r = d['roofline']
print(d['config']['workload'][:60], d['ms_per_step'], d['value'], d['residual'], d.get('ms_per_step_scheduler_in_loop'), d.get('gstrs_s'), d['roofline']['model_T_star_over_t_gstrf'])
print({k: r.get(k) for k in ('kernel', 'achieved', 'frac', 'traffic', 'traffic_over_algorithmic', 'traffic_note')})
print({k:(v['ms'],v['launches']) for k,v in d['kernels'].items()})
PY
head -10 gpurun_out/r04al_shell398_table.md | cut -c1-200
head -12 gpurun_out/r04al_shell398_critical_path.md | cut -c1-200
