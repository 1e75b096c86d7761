#!/bin/bash
# round 4, very last build (alignment sizes count the vertices taken out before the dissection): the kkt cases of the GPU suite, the profile recipe on both
# bench workloads (-> profiles/hbm_traffic.json), the default line as the driver runs it, and the nlpkkt class lines.  (The whole GPU suite ran on the build
# before: profiles/r04an_gputests_before_alignment_fix.log; the two differ in the ordering of matrices with constraint rows only.)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( time timeout 300 python -m pytest tests -m gpu -x -q -k "kkt" ) > gpurun_out/r04ao_gputests_kkt.log 2>&1
tail -3 gpurun_out/r04ao_gputests_kkt.log
bash tools/profile_recipe.sh r04ao_elastic3d_77 > gpurun_out/r04ao_profile_recipe.log 2>&1
bash tools/profile_recipe.sh r04ao_shell398 --workload shell >> gpurun_out/r04ao_profile_recipe.log 2>&1
cp gpurun_out/hbm_traffic.json profiles/hbm_traffic.json 2>/dev/null
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r04ao_bench_default.log 2> gpurun_out/r04ao_bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04ao_bench_default.log').readline())
r = d['roofline']
print(d['config']['workload'][:70], d['ms_per_step'], d['value'], d['residual'], d['factor_check'], d.get('ms_per_step_scheduler_in_loop'), d.get('gstrs_s'))
print({k: r.get(k) for k in ('achieved', 'frac', 'traffic', 'traffic_over_algorithmic', 'traffic_note', 'mfma_executed_tflops', 'avg_launch_us')})
print("cpu", d['cpu_baseline'] and d['cpu_baseline']['value'])
print([(s['workload'][:22], round(s['ms_per_step'],2), round(s['value']), s['residual']) for s in d.get('secondary') or []])
PY
for w in "kkt --size 56" "kkt --size 120"; do
  line=$(timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-profile-pass --no-secondary --no-sched-steps 2>/dev/null | tail -1)
  echo "R64 $w :: $(python -c "import json,sys; d=json.loads(sys.argv[1]); c=d['config']; print('n %d  nnz %d  F %.3e  %.2f ms  %.2f TFLOP/s  residual %.2e  factor check %.2e  gstrs %.3f s  init %.1f s' % (c['n'], c['nnz'], c['flop'], d['ms_per_step'], d['value']/1e3, d['residual'], d['factor_check'], d['gstrs_s'], d['init_s']))" "$line" 2>&1 | tail -1)" | tee -a gpurun_out/r04ao_kkt_class.log
done
