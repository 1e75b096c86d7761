set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for la in 2 4 16 32; do
PANGULU_AMD_LOOKAHEAD_MAX_GETRF=$la timeout 300 python tools/sweep_opt.py 11 512 2>&1 | tail -1
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1g -o runc -- python3 $GRAFT_REPO_ROOT/tools/sweep_opt.py 11 512 2>&1 | tail -2
