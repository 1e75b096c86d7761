cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1w -o tl -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-profile-pass --steps 1 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_r1w.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_r1w -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f > gpurun_out/timeline_r1w.txt 2>&1
rm -f $f
