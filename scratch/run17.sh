cd $GRAFT_REPO_ROOT
for w in -1 0 9 64; do
PANGULU_AMD_EAGER_UPDATES_MIN_GETRF=$w timeout 300 python tools/sweep_opt.py 2 10 2>&1 | tail -1
done
