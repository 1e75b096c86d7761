cd $GRAFT_REPO_ROOT
for d in 1 0; do
PANGULU_HIP_OCCUPANCY_SUMMARIES=$d PANGULU_HIP_DEBUG_SSSSM=1 PANGULU_HIP_HOST_TIMING=1 timeout 300 python tools/sweep_opt.py 2 10 2>&1 | grep "stamps\|host seconds\|option" | tail -4 | cut -c1-330
done
