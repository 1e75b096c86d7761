cd $GRAFT_REPO_ROOT/tools/microbench
for nd in 1024 256 64 16; do for nops in 1500 100; do
 echo "== distinct destinations $nd operands $nops"
 ./bench_dense_w2.bin 4096 3 1 16 $nd $nops | tail -1
done; done
