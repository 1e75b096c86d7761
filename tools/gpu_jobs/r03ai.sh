#!/bin/bash
# sanity of the clean-built libraries: smoke, a slice of the GPU suite, the shell line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( timeout 900 python -m pytest tests/test_gpu_smoke_bench.py tests/test_gpu_parity.py tests/test_update_values.py -m gpu -q -x ) 2>&1 | tail -3
timeout 600 python bench.py --workload shell --steps 20 --warmup 3 2>/dev/null | cut -c1-200
