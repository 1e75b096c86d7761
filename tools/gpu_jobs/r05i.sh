#!/bin/bash
# round 5: the new parity cases on the GPU -- seeded random switch sweep, replay with poisoned receive slots, golden operator
# vectors under their frozen permutation, the KKT class on its quasi-definite matrix
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests/test_gpu_env_switches.py -m gpu -q -k "random_switch_sweep" -x --durations=5 ) > gpurun_out/r05i_sweep.log 2>&1
tail -12 gpurun_out/r05i_sweep.log
( timeout 1500 python -m pytest tests/test_multirank.py tests/test_gpu_operators.py tests/test_gpu_parity_scale.py -m gpu -q -k "replay or committed_vectors or kkt or saddle" --durations=5 ) > gpurun_out/r05i_other.log 2>&1
tail -12 gpurun_out/r05i_other.log
