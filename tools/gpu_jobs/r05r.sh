#!/bin/bash
# round 5: rocprofv3 kernel statistics + PMC passes of the final build on the headline workload, the KKT class line, lone panel solves
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05r}
bash tools/gpu_jobs/r05q.sh
( timeout 1200 python bench.py --workload kkt --size 120 --steps 5 --warmup 2 --no-secondary ) > gpurun_out/${TAG}_bench_kkt120.json.log 2> gpurun_out/${TAG}_bench_kkt120.err
echo "kkt bench rc $?"
python tools/ab_summary.py gpurun_out/${TAG}_bench_kkt120.json.log
timeout 2400 tools/profile_recipe.sh ${TAG}_elastic3d_77
ls -la gpurun_out/ | tail -12
