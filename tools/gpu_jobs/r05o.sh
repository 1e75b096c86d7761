#!/bin/bash
# round 5: stand-alone GETRF microbench (band chunk 3 against 4) + in-situ A/B with the current library
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TAG=${TAG:-r05o} bash tools/gpu_jobs/r05n.sh
TAG=${TAG:-r05o} bash tools/gpu_jobs/r05l.sh
