cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
timeout 300 python tools/sweep_opt.py 2 10 10 2>&1 | tail -2
./tools/microbench/bench_dense.bin 4096 3 1 16 | tail -1
./tools/microbench/bench_dense.bin 4096 3 1 8 | tail -1
