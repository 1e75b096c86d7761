cd $GRAFT_REPO_ROOT/tools/microbench
export HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "3.25 3.25 1" "3.25 12 6" "3.25 60 8" "1.7 60 8"; do echo "== $cfg"; timeout 25 ./ipc_probe.bin $cfg; echo "rc $?"; done
