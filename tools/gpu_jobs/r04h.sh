#!/bin/bash
# round 4: (1) the RCCL data plane executed on the one-GPU box (ranks made to look like different hosts: RCCL's socket transport);
# (2) the default line as the driver runs it: elastic3d(77) + secondary shell(398), fem27(112) + cpu_baseline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( time NCCL_DEBUG=WARN timeout 900 python -m pytest tests/test_multirank.py -m gpu -x -q -k "rccl" ) > gpurun_out/r04h_rccl_tests.log 2>&1
tail -30 gpurun_out/r04h_rccl_tests.log | cut -c1-300
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r04h_bench_default.log 2> gpurun_out/r04h_bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04h_bench_default.log').readline())
print(d['config']['workload'][:60], d['ms_per_step'], d['value'], d['residual'], d['factor_check'], d.get('ms_per_step_scheduler_in_loop'), d.get('gstrs_s'), d['hbm_used_GB'])
print("roofline", d['roofline']['frac'], d['roofline'].get('traffic'), "cpu", d['cpu_baseline'] and (d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:80]))
print([(s['workload'][:22], round(s['ms_per_step'],2), round(s['value']), s['residual']) for s in d.get('secondary') or []])
PY
tail -4 gpurun_out/r04h_bench_default.err | cut -c1-200
