"""bench.py's supervisors on the CPU (no GPU, no solver): the walk through the transports, the time budget and the transport A/B, with
workers that bench.py fakes under PANGULU_BENCH_TEST_FAKE_WORKER -- what a first run on eight GPUs depends on when a transport passes its
start-up self-test and then stalls.  The same flow with real workers is tests/test_gpu_smoke_bench.py."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(fake, *extra, timeout=300):
    env = dict(os.environ, PANGULU_BENCH_TEST_FAKE_WORKER=fake)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + list(extra),
                         capture_output=True, text=True, timeout=timeout, env=env)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    return out, (json.loads(lines[-1]) if lines else None), time.time() - t0


def test_first_transport_runs_and_the_other_one_is_measured_beside_it():
    out, line, _ = run("rccl=ok,ipc=ok")
    assert out.returncode == 0 and line is not None, out.stderr[-2000:]
    att = line["config"]["worker_attempts"]
    assert [a["transport"] for a in att] == ["rccl"] and att[0]["done"] and att[0]["cap_s"] <= 400
    ab = line["transport_ab"]
    assert ab["transport"] == "ipc" and ab["headline_transport"] == "rccl" and ab["ms_per_step"] == 1.0 and ab["steps"] == 2
    assert line["bench_wall_s"] < line["time_budget_s"] == 1500


def test_a_transport_that_is_not_there_and_one_that_hangs():
    """rccl: not available (the worker ends by itself) -> ipc: hangs, every supervisor gives it up at its cap -> host: runs.  No A/B
    behind a first attempt that failed."""
    out, line, wall = run("rccl=fail,ipc=hang,host=ok", "--worker-timeout", "8")
    assert out.returncode == 0 and line is not None, out.stderr[-2000:]
    att = line["config"]["worker_attempts"]
    assert [a["transport"] for a in att] == ["rccl", "ipc", "host"], att
    assert not att[0]["done"] and not att[1]["done"] and 7 <= att[1]["s"] <= 20 and att[1]["cap_s"] == 8 and att[2]["done"]
    assert line["transport_ab"] is None and line["config"]["transport"] == "host" and wall < 120


def test_every_transport_hangs_inside_the_budget():
    """Nothing works: three attempts, each given up at its cap, a non-zero exit and no line -- in bounded time."""
    out, line, wall = run("rccl=hang,ipc=hang,host=hang", "--worker-timeout", "5", "--total-budget", "60")
    assert out.returncode != 0 and line is None and wall < 60


def test_the_ab_failing_changes_nothing_but_its_object():
    out, line, _ = run("rccl=ok,ipc=hang", "--total-budget", "75")  # (the A/B gets what is left: about a minute)
    assert out.returncode == 0 and line is not None, out.stderr[-2000:]
    assert line["value"] == 1.0 and "error" in line["transport_ab"] and line["transport_ab"]["transport"] == "ipc"
