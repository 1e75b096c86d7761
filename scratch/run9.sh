cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.log; cut -c1-400 gpurun_out/bench_default.log
