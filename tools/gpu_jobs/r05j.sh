#!/bin/bash
# round 5: the whole GPU suite
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 ) > gpurun_out/${TAG:-r05j}_gputests.log 2>&1
tail -25 gpurun_out/${TAG:-r05j}_gputests.log
