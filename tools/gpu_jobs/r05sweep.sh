#!/bin/bash
# round 5: the random switch sweep with two other seeds, 80 draws each (the suite runs seed 20261003, 60 draws)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
TAG=${TAG:-r05sweep}
for seed in 777 31337; do
( PG_SWEEP_SEED=$seed PG_SWEEP_DRAWS=80 timeout 1500 python -m pytest tests/test_gpu_env_switches.py -m gpu -q -k "random_switch_sweep" ) > gpurun_out/${TAG}_seed$seed.log 2>&1
tail -4 gpurun_out/${TAG}_seed$seed.log
done
