cd $GRAFT_REPO_ROOT
for w in 16384 4096 65536; do
PANGULU_HIP_MIRROR_JOB_WGS=$w timeout 300 python tools/sweep_opt.py 2 10 2>&1 | tail -1
done
