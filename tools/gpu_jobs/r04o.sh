#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( time PANGULU_TEST_SHOW_OUTPUT=1 PANGULU_AMD_TRACE=1 PANGULU_TEST_RANK_TIMEOUT=300 timeout 900 python -m pytest tests/test_multirank.py -m gpu -x -q -s -k "replay and shell_40x40 and 2-" ) > gpurun_out/r04o_replay_debug.log 2>&1
grep -E "trace\] rank|log of|replay|passed|failed|Error|error" gpurun_out/r04o_replay_debug.log | cut -c1-260 | head -60
