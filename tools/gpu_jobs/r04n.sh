#!/bin/bash
# round 4: multi-rank replay of a rank's log (PANGULU_AMD_MULTI_REPLAY=1) on ranks sharing the GPU, then the whole multi-rank suite
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( time PANGULU_AMD_TRACE=1 PANGULU_TEST_RANK_TIMEOUT=300 timeout 1500 python -m pytest tests/test_multirank.py -m gpu -x -q -k "replay" ) > gpurun_out/r04n_replay_tests.log 2>&1
tail -40 gpurun_out/r04n_replay_tests.log | cut -c1-300
( time timeout 1500 python -m pytest tests/test_multirank.py -m gpu -x -q -k "not replay" ) > gpurun_out/r04n_multirank_tests.log 2>&1
tail -4 gpurun_out/r04n_multirank_tests.log
