#!/bin/bash
# round 4, final tree: the whole GPU suite, the profile recipe on the default workload (rocprofv3 stats + PMC passes -> profiles/hbm_traffic.json),
# then the default line exactly as the driver runs it
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/r04an_gputests.log 2>&1
tail -4 gpurun_out/r04an_gputests.log
bash tools/profile_recipe.sh r04an_elastic3d_77 > gpurun_out/r04an_profile_recipe.log 2>&1
tail -5 gpurun_out/r04an_profile_recipe.log | cut -c1-200
cp gpurun_out/hbm_traffic.json profiles/hbm_traffic.json 2>/dev/null
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r04an_bench_default.log 2> gpurun_out/r04an_bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04an_bench_default.log').readline())
r = d['roofline']
print(d['config']['workload'][:70], d['ms_per_step'], d['value'], d['residual'], d['factor_check'], d.get('ms_per_step_scheduler_in_loop'), d.get('gstrs_s'))
print({k: r.get(k) for k in ('achieved', 'frac', 'traffic', 'traffic_over_algorithmic', 'traffic_note', 'mfma_executed_tflops', 'avg_launch_us')})
print("cpu", d['cpu_baseline'] and (d['cpu_baseline']['value'], d['cpu_baseline']['sample'][-200:]))
print([(s['workload'][:22], round(s['ms_per_step'],2), round(s['value']), s['residual']) for s in d.get('secondary') or []])
PY
head -12 gpurun_out/r04an_elastic3d_77_table.md | cut -c1-220
# ... and the complex class on the same build
for N in 48 64 80 96; do timeout 900 python tools/bench_cr64.py $N 128 2>&1 | grep "permille=2:" | tee -a gpurun_out/r04an_cr64.log; done
# ... and both Serena-class stand-ins without coordinates (graph-only ordering of this build)
for w in "elastic3d" "fem27"; do
( time timeout 1200 python bench.py --workload $w --no-coords --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-sched-steps --no-profile-pass ) > gpurun_out/r04an_${w}_nocoords.log 2> gpurun_out/r04an_${w}_nocoords.err
python - <<PY
import json
d = json.loads(open('gpurun_out/r04an_${w}_nocoords.log').readline())
c = d['config']
print(c['workload'][:60], 'F %.3e' % c['flop'], d['ms_per_step'], d['value'], d['residual'], 'init', d['init_s'])
PY
done
# ... and the nlpkkt class at its full order (constraint rows eliminated ahead of their unknowns)
for w in "kkt --size 56" "kkt --size 120" "kkt --size 120 --no-coords"; do
  line=$(timeout 1200 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-profile-pass --no-secondary --no-sched-steps 2>/dev/null | tail -1)
  echo "R64 $w :: $(python -c "import json,sys; d=json.loads(sys.argv[1]); c=d['config']; print('n %d  nnz %d  F %.3e  %.2f ms  %.2f TFLOP/s  residual %.2e  factor check %.2e  gstrs %.3f s  init %.1f s  hbm %s' % (c['n'], c['nnz'], c['flop'], d['ms_per_step'], d['value']/1e3, d['residual'], d['factor_check'], d['gstrs_s'], d['init_s'], d.get('hbm_used_GB')))" "$line" 2>&1 | tail -1)" | tee -a gpurun_out/r04an_kkt_class.log
done
