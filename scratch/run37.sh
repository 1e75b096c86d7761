R=$GRAFT_REPO_ROOT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-profile-pass --steps 2 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1y -o runc -- $B 2>&1 | grep metric | cut -c1-160
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_r1y_fetch -o runc -- $B 2>&1 | grep metric | cut -c1-120
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_r1y_write -o runc -- $B 2>&1 | grep metric | cut -c1-120
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/prof_r1y_sq -o runc -- $B 2>&1 | grep metric | cut -c1-120
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof_r1y_mfma -o runc -- $B 2>&1 | grep -i "metric" | cut -c1-120
cd $R
find gpurun_out/prof_r1y* -name "*kernel_trace.csv" -delete
python tools/summarize_rocprof.py gpurun_out/prof_r1y gpurun_out/prof_r1y_fetch gpurun_out/prof_r1y_write gpurun_out/prof_r1y_sq gpurun_out/prof_r1y_mfma --json gpurun_out/r01y_hbm_traffic_shell398.json > gpurun_out/r01y_table.md
cp gpurun_out/r01y_hbm_traffic_shell398.json profiles/
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.log; cut -c1-300 gpurun_out/bench_default.log
