cd $GRAFT_REPO_ROOT/tools/microbench
echo "--- MFMA update, all tiles live: groups tasks/group atomic"
./bench_dense.bin 4096 8 0 | tail -1
./bench_dense.bin 4096 1 0 | tail -1
./bench_dense.bin 4096 8 0 4 | tail -1
echo "--- dense solves, all tiles live: tasks tstrf distinctLU direct"
./bench_trsm.bin 4096 1 64 1 | tail -1
./bench_trsm.bin 4096 1 64 0 | tail -1
./bench_trsm.bin 4096 0 64 1 | tail -1
./bench_trsm.bin 4 1 1 1 | tail -1
./bench_trsm.bin 4 1 1 0 | tail -1
