#!/bin/bash
# round 4: orderings on the device -- graph-only dissection (bench.py --no-coords) with the separator-internal order variants,
# and the 13-direction geometric cuts on the 7-point class
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
OUT=gpurun_out/r04b_orderings.log
: > $OUT
run() { # label, env..., -- bench args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  line=$(env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-profile-pass --steps 4 --warmup 1 "$@" 2>gpurun_out/r04b_last.err | tail -1)
  echo "$label :: $(python - "$line" <<'PY'
import json,sys
try:
    d=json.loads(sys.argv[1]); c=d["config"]
    print("ms %.2f  TFLOP/s %.2f  F %.4e  fill %.1fM  residual %.2e  init %.1fs  ordering: %s" % (d["ms_per_step"], d["value"]/1e3, c["flop"], c["symbolic_nnz"]/1e6, d["residual"], d.get("init_s",0), c["ordering"][:60]))
except Exception as e:
    print("FAILED", e, sys.argv[1][:300])
PY
)" | tee -a $OUT
}
for mode in kd surface natural; do
  run "fem27(80) no-coords sep-order=$mode" PANGULU_AMD_SEPARATOR_ORDER_GRAPH=$mode -- --workload fem27 --size 80 --no-coords
done
run "fem27(80) coords" A=1 -- --workload fem27 --size 80
for mode in kd natural; do
  run "shell(300) no-coords sep-order=$mode" PANGULU_AMD_SEPARATOR_ORDER_GRAPH=$mode -- --workload shell --size 300 300 --no-coords
done
run "shell(300) coords" A=1 -- --workload shell --size 300 300
for mode in kd natural; do
  run "poisson3d(80) no-coords sep-order=$mode" PANGULU_AMD_SEPARATOR_ORDER_GRAPH=$mode -- --workload poisson --size 80 --no-coords
done
run "poisson3d(80) coords (13 directions)" A=1 -- --workload poisson --size 80
run "poisson3d(80) coords (axes only)" PANGULU_AMD_ND_DIAGONALS=0 -- --workload poisson --size 80
run "poisson3d(96) coords (13 directions)" A=1 -- --workload poisson --size 96
for mode in kd natural; do
  run "fem27(112) no-coords sep-order=$mode" PANGULU_AMD_SEPARATOR_ORDER_GRAPH=$mode -- --workload fem27 --no-coords
done
