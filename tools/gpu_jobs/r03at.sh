#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
( timeout 900 python -m pytest tests/test_gpu_env_switches.py -m gpu -q -x -k "SEPARATOR_ORDER or HEAVY_FIRST or CHASE or defaults" ) 2>&1 | tail -3
