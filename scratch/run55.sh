cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
PANGULU_HIP_TRSM_DIRECT=1 timeout 900 python -m pytest tests/test_multirank.py -x -q -m gpu 2>&1 | tail -40 | cut -c1-300
