#!/bin/bash
# round 5: profile recipe (kernel statistics + PMC passes) of the current build on the headline workload
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05x}
timeout 2400 tools/profile_recipe.sh ${TAG}_elastic3d_77
cat gpurun_out/${TAG}_elastic3d_77_critical_path.md
