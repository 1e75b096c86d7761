#!/bin/bash
# round 5: the final build on the other classes: profile of shell(398) (ldoor class), bench lines of kkt(120) and complex Poisson 64^3
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05fin}
timeout 1200 tools/profile_recipe.sh ${TAG}_shell398 --workload shell --size 398 398
cat gpurun_out/${TAG}_shell398_critical_path.md
( timeout 1200 python bench.py --workload kkt --size 120 --steps 5 --warmup 2 --no-secondary ) > gpurun_out/${TAG}_bench_kkt120.json.log 2> gpurun_out/${TAG}_bench_kkt120.err
echo "kkt bench rc $?"
python tools/ab_summary.py gpurun_out/${TAG}_bench_kkt120.json.log
