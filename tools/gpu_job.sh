#!/bin/bash
# The ONE script for work on the GPU box (replaces 111 one-off tools/gpu_jobs/*.sh of rounds 3-5; they are in the history):
#
#   gpurun --timeout S -- 'tools/gpu_job.sh <job> [args]; tools/gpu_job.sh <job> [args]; ...'
#
#   tests [pytest args]              the GPU suite (default: tests -m gpu -x -q --durations=15)      -> gpurun_out/<TAG>_gputests.log
#   bench [bench.py flags]           one bench.py run                                                -> gpurun_out/<TAG>_bench.json.log / .err
#   ab "<ENV=a ENV2=b>|..." [flags]  A/B of environment switches on one box: one bench.py run per '|'-separated setting ("-" = none),
#                                    --no-cpu-baseline --no-secondary --no-sched-steps --no-profile-pass added; prints ms_per_step each
#   profile <tag> [bench.py flags]   tools/profile_recipe.sh: rocprofv3 kernel stats + the PMC passes -> gpurun_out/<tag>_table.md, hbm_traffic.json
#   micro <source.hip> [args]        build tools/microbench/<source.hip> on the box with hipcc and run it -> gpurun_out/<TAG>_micro.log
#   repeat <n> <pytest node id>      the same test n times, stop at the first failure (start-up races) -> gpurun_out/<TAG>_repeat.log
#   launchlog [bench.py flags]       per-launch log of the profiled factorisation + tools/launch_log_summary.py
#
# TAG (environment, default "job") names the outputs.  Everything is written under gpurun_out/ (scratch; copy what is to be judged
# into profiles/ and add its line to profiles/INDEX.md).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-job}
JOB=${1:-help}
shift || true
QUIET="--no-cpu-baseline --no-secondary --no-sched-steps --no-profile-pass"

ms_of() { python3 - "$1" <<'EOF'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        j = json.loads(ln)
        k = j.get("kernels") or {}
        print("ms_per_step %.2f  value %s  residual %s  update class ms %s" % (j["ms_per_step"], j["value"], j["residual"], (k.get("ssssm_dense_mfma") or {}).get("ms")))
        break
else:
    print("no metric line")
EOF
}

case "$JOB" in
tests)
    if [ $# -eq 0 ]; then set -- tests -m gpu -x -q --durations=15; fi
    timeout ${TEST_TIMEOUT:-1500} python3 -m pytest "$@" > gpurun_out/${TAG}_gputests.log 2>&1
    echo "[gpu_job tests] rc=$?"; tail -25 gpurun_out/${TAG}_gputests.log
    ;;
bench)
    timeout ${BENCH_TIMEOUT:-1700} python3 bench.py "$@" > gpurun_out/${TAG}_bench.json.log 2> gpurun_out/${TAG}_bench.err
    echo "[gpu_job bench] rc=$?"; ms_of gpurun_out/${TAG}_bench.json.log; tail -3 gpurun_out/${TAG}_bench.err
    ;;
ab)
    SETTINGS=$1; shift
    IFS='|' read -ra LIST <<< "$SETTINGS"
    i=0
    for s in "${LIST[@]}"; do
        [ "$s" = "-" ] && s=""
        out=gpurun_out/${TAG}_ab${i}.json.log
        env $s timeout ${BENCH_TIMEOUT:-900} python3 bench.py $QUIET "$@" > $out 2> gpurun_out/${TAG}_ab${i}.err
        echo "[gpu_job ab $i] '${s:-default}' rc=$? $(ms_of $out)" | tee -a gpurun_out/${TAG}_ab_summary.txt
        i=$((i + 1))
    done
    ;;
profile)
    T=${1:-$TAG}; shift || true
    tools/profile_recipe.sh "$T" "$@"
    cat gpurun_out/${T}_table.md | head -30
    ;;
micro)
    SRC=$1; shift
    BIN=/tmp/$(basename "$SRC" .hip)_$(echo "${MICRO_FLAGS:-}" | md5sum | cut -c1-8).bin
    if [ ! -x $BIN ]; then  # (one build per source and flag set on a box: several runs of one call share it)
        /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics ${MICRO_FLAGS:-} -I pangulu_amd/csrc/platform -I include tools/microbench/$SRC -o $BIN 2>&1 | tail -5
    fi
    timeout ${MICRO_TIMEOUT:-300} $BIN "$@" > gpurun_out/${TAG}_micro.log 2>&1
    echo "[gpu_job micro] rc=$?"; tail -40 gpurun_out/${TAG}_micro.log
    ;;
repeat)
    N=$1; shift
    : > gpurun_out/${TAG}_repeat.log
    for i in $(seq 1 "$N"); do
        if ! timeout 600 python3 -m pytest "$@" -x -q >> gpurun_out/${TAG}_repeat.log 2>&1; then
            echo "[gpu_job repeat] FAILED at repetition $i of $N" | tee -a gpurun_out/${TAG}_repeat.log
            tail -60 gpurun_out/${TAG}_repeat.log
            exit 1
        fi
    done
    echo "[gpu_job repeat] $N of $N passed" | tee -a gpurun_out/${TAG}_repeat.log
    ;;
launchlog)
    PANGULU_HIP_LAUNCH_LOG=$R/gpurun_out/${TAG}_launch_log.txt timeout 900 python3 bench.py --no-cpu-baseline --no-secondary --no-sched-steps --steps 2 --warmup 1 "$@" \
        > gpurun_out/${TAG}_bench.json.log 2> gpurun_out/${TAG}_bench.err
    python3 tools/launch_log_summary.py gpurun_out/${TAG}_launch_log.txt | tee gpurun_out/${TAG}_launch_log_summary.txt
    gzip -f gpurun_out/${TAG}_launch_log.txt
    ;;
*)
    sed -n 2,18p "$0"
    ;;
esac
