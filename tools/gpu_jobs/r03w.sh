#!/bin/bash
# where the ldoor-class factorisation spends its time now: profile recipe on shell(398)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
bash tools/profile_recipe.sh r03w_shell398 --workload shell > gpurun_out/r03w_profile_recipe.log 2>&1
cat gpurun_out/r03w_shell398_table.md | cut -c1-230 | head -18; cat gpurun_out/r03w_shell398_critical_path.md | head -24
