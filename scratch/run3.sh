cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1n -o runc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-profile-pass --steps 2 --warmup 1 2>&1 | grep -v "^W2026\|^E2026" | tail -3
