#!/bin/bash
# round 3, final checkpoint of the round: + separators in k-d order -- GPU suite, bench lines, two self-launched
# ranks, profile recipe (kernel stats + PMC passes) for the default workload
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1800 python -m pytest tests -m gpu -q ) > gpurun_out/r03ao_pytest.log 2>&1
tail -8 gpurun_out/r03ao_pytest.log
( time timeout 1500 python bench.py ) > gpurun_out/r03ao_bench_default.log 2> gpurun_out/r03ao_bench_default.err
cut -c1-700 gpurun_out/r03ao_bench_default.log; tail -4 gpurun_out/r03ao_bench_default.err
( time timeout 900 python bench.py --workload shell --steps 20 --warmup 3 ) > gpurun_out/r03ao_bench_shell.log 2> gpurun_out/r03ao_bench_shell.err
cut -c1-500 gpurun_out/r03ao_bench_shell.log; tail -3 gpurun_out/r03ao_bench_shell.err
( time timeout 900 python bench.py --gpus 2 --workload shell --steps 5 --warmup 2 ) > gpurun_out/r03ao_bench_shell_n2.log 2> gpurun_out/r03ao_bench_shell_n2.err
cut -c1-900 gpurun_out/r03ao_bench_shell_n2.log; tail -5 gpurun_out/r03ao_bench_shell_n2.err
bash tools/profile_recipe.sh r03ao_fem27_112 > gpurun_out/r03ao_profile_recipe.log 2>&1
cat gpurun_out/r03ao_fem27_112_table.md | cut -c1-220 | head -16; cat gpurun_out/r03ao_fem27_112_critical_path.md | head -20
ls -la gpurun_out/hbm_traffic.json
