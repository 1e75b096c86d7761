"""The driver's entry points on the GPU: smoke() and a tiny bench.py run must produce a valid JSON line."""
import json
import os
import subprocess
import sys

import pytest

from .helpers import ROOT

pytestmark = pytest.mark.gpu


def test_smoke():
    import __graft_entry__

    __graft_entry__.smoke()


def run_bench(*extra, timeout=1500):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "shell", "--size", "40", "40",
                          "--steps", "2", "--warmup", "1"] + list(extra), capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "residual", "factor_check", "model")


def test_bench_json_contract():
    line = run_bench("--cpu-sample-stride", "3")
    for key in CONTRACT_KEYS:
        assert key in line, key
    assert line["unit"] == "GFLOP/s" and line["dtype"] == "f64" and line["n_gpus"] == 1 and line["steps"] == 2
    # both correctness criteria of the reference, from the last TIMED step (timed configuration: COUNT_FLOPS off)
    assert line["value"] > 0 and line["residual"] < 1e-10 and line["factor_check"] < 1e-12
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    cpu = line["cpu_baseline"]
    assert set(cpu) >= {"value", "unit", "cores", "kind", "sample", "cpu_model"}
    assert "every 3rd task" in cpu["sample"] and cpu["value"] > 0 and cpu["cores"] == 1 and cpu["wall_value"] <= cpu["value"]
    # mirror maintenance is part of the reported kernel time, and the whole-factorisation bound T* is there
    assert {"densify", "sparsify"} & set(line["kernels"]) or line["config"]["nb"] != 256
    assert 0 < line["roofline"]["model_T_star_over_t_gstrf"] < 1
    assert line["model"]["rank_flop_share_max_over_mean"] == 1.0 and line["model"]["critical_path_tasks"] > 0
    # the measurement row adds up (a solve launch was booked under one class and GESSM fell out of `kernels`):
    # every task class is listed, and their structural flops sum to the factorisation's
    k = line["kernels"]
    assert {"getrf", "tstrf", "gessm"} <= set(k), sorted(k)
    assert k["tstrf"]["tasks"] == line["config"]["tasks_rank0"]["tstrf"] and k["gessm"]["tasks"] == line["config"]["tasks_rank0"]["gessm"]
    total = sum(k[c]["GFLOP"] for c in ("getrf", "tstrf", "gessm", "ssssm_sparse", "ssssm_dense_mfma") if c in k)
    assert abs(total - line["config"]["flop"] / 1e9) <= 1e-6 * line["config"]["flop"] / 1e9 + 1e-3, (total, line["config"]["flop"])
    per = line["roofline"]["per_kernel"]
    assert {"trsm", "getrf"} <= set(per) and per["getrf"]["ms"] > 0 and per["trsm"]["ms"] > 0
    if "ssssm_class" in per:
        assert per["ssssm_general"]["ms"] + per["ssssm_front"]["ms"] <= per["ssssm_class"]["ms"] * 1.001
        assert per["ssssm_class"]["executed_over_structural"] >= 1.0
    assert line["ms_per_step_first_factorisation"] > 0 and line["bench_wall_s"] < line["time_budget_s"]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: fresh rank processes, a complete line.  On a one-GPU box both
    ranks share the device: RCCL refuses that, the transports fall back together and the line says so."""
    line = run_bench("--gpus", "2", "--cpu-sample-stride", "3", "--transport", "ipc")
    for key in CONTRACT_KEYS:
        assert key in line, key
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["residual"] < 1e-10 and line["factor_check"] < 1e-12
    cfg = line["config"]
    assert cfg["transport"] in ("host", "ipc", "rccl") and len(cfg["transport_tried"]) >= 1
    assert cfg["rccl_nranks"] == (2 if cfg["transport"] == "rccl" else 0)
    cpu = line["cpu_baseline"]
    assert cpu["cores"] == 1 and cpu["value"] > 0, cpu   # (N > 1: one rank x one thread for reference; the contract's CPU baseline is the N = 1 line's)
    assert line["ms_per_step_first_factorisation"] > 0
    # the other device transport, three steps: an object with a number or with the reason there is none (one GPU: RCCL refuses two ranks
    # on one device, so the headline ran on ipc and the A/B's rccl worker ends as "not available")
    ab = line["transport_ab"]
    assert ab is not None and ab["transport"] in ("rccl", "host") and ("error" in ab or ab["ms_per_step"] > 0), ab
    m = line["model"]
    # (T*(N) = max over ranks of T*_r + link term: at this size the link term is the larger part)
    assert m["T_star_ms"] >= m["sum_over_ranks_ms"] / 2 and 1.0 <= m["rank_flop_share_max_over_mean"] < 2.0
    assert m["sent_GB"] > 0 and m["link_term_ms_max"] > 0


def test_bench_on_a_matrix_file_without_coordinates(tmp_path):
    """`bench.py --mtx`: a MatrixMarket file and the reference's binary .lid (examples/example.c:112-163), as the real BASELINE
    matrices would arrive -- no coordinates, so the graph-only ordering runs -- plus a right-hand side file; and `--no-coords` on
    a generator.  Residual below 1e-10, the ordering named in the line, new keys of round 4 present."""
    import scipy.io

    from pangulu_amd import matrices as M

    n, cp, ri, va, _ = M.fem27(14)
    A = M.to_scipy(n, cp, ri, va)
    mtx = str(tmp_path / "fem27_14.mtx")
    scipy.io.mmwrite(mtx, A)
    lid = str(tmp_path / "fem27_14.lid")
    M.write_lid(lid, n, cp, ri, va)
    rhs = str(tmp_path / "b.rhs")
    b = M.rhs_of_ones(n, cp, ri, va)
    with open(rhs, "w") as f:
        f.write("%d\n" % n)
        f.write("\n".join("%.17g" % x for x in b) + "\n")
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    flops = []
    for extra in (["--mtx", mtx], ["--mtx", lid, "--rhs", rhs], ["--workload", "fem27", "--size", "14", "--no-coords"]):
        out = subprocess.run(base + extra, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
        line = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith('{"metric"')][-1])
        assert line["residual"] < 1e-10 and line["factor_check"] < 1e-12 and line["value"] > 0
        assert "graph only" in line["config"]["ordering"], line["config"]["ordering"]
        assert line["gstrs_s"] > 0 and "hbm_breakdown_GB" in line and line["ms_per_step_scheduler_in_loop"] > 0
        assert line["data"] == ("file" if "--mtx" in extra else "synthetic")
        flops.append(line["config"]["flop"])
    assert flops[0] == flops[1] == flops[2]  # the same matrix three ways: the same ordering, the same structural flops


def test_bench_eight_ranks_on_one_gpu_and_multi_replay():
    """First-contact rehearsal of the driver's 8-GPU run on the one GPU of the box (VERDICT r4 next #3c): `bench.py --gpus 8`
    self-launched -- eight supervisors, the transport order rccl -> ipc -> host walked by all of them together (RCCL refuses eight
    ranks on one device), cpu_baseline R x 1 on request (--cpu-ranks-leg) -- once with the scheduler in every step (--no-multi-replay) and once in the default mode (every rank
    replays the log of its first factorisation).  Eight ranks share 288 GB here, so the matrix is small; what is checked is the contract, not a rate."""
    for extra in (["--no-multi-replay"], []):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "fem27", "--size", "32", "--nb", "128",
                              "--steps", "3", "--warmup", "2", "--cpu-sample-stride", "4"] + (extra or ["--cpu-ranks-leg"]), capture_output=True, text=True, timeout=1500)
        assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
        line = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith('{"metric"')][-1])
        for key in CONTRACT_KEYS:
            assert key in line, key
        assert line["n_gpus"] == 8 and line["value"] > 0 and not line["parity_failed"]
        assert line["residual"] < 1e-10 and line["factor_check"] < 1e-10
        cfg = line["config"]
        assert cfg["transport"] in ("ipc", "host") and cfg["transport_tried"], cfg  # (one device: no RCCL communicator over eight ranks)
        assert any(a["transport"] == "rccl" for a in cfg["worker_attempts"]) and cfg["worker_attempts"][-1]["done"], cfg["worker_attempts"]
        if extra:
            assert line["cpu_baseline"]["cores"] == 1 and "one_rank_x_one_thread" not in line["cpu_baseline"]
        else:
            assert line["cpu_baseline"]["cores"] == 8 and line["cpu_baseline"]["one_rank_x_one_thread"]["cores"] == 1
        assert line["bench_wall_s"] < line["time_budget_s"] and all("cap_s" in a for a in cfg["worker_attempts"])
        m = line["model"]
        assert m["sent_GB"] > 0 and m["latency_chain_ms"] > m["critical_path_ms"] and 1.0 <= m["rank_flop_share_max_over_mean"] < 3.0
        assert m["hbm_fullest_rank_GB"]["total"] > 0
        # (the default: every rank replays the log of its first factorisation; ipc defers sends, host staging cannot be replayed)
        assert line["static_schedule_replayed"] is (not extra and cfg["transport"] == "ipc"), (extra, cfg["transport"], line["static_schedule_replayed"])


def test_bench_scaling_prediction_and_parity_gate():
    """One rank: the line carries the structure's forecast for 1 / 2 / 4 / 8 ranks (T*(N), link term, latency-aware chain, HBM of
    the fullest rank).  And THE GATE: a factorisation whose factors fail the reference's criteria must not publish a rate -- forced
    here with PANGULU_BENCH_TEST_BREAK_FACTORS (bench.py takes the residual against a matrix whose largest entry was changed behind the
    factorisation's back, bench.py:gpu_worker_main -- the factors are then those of the wrong matrix): value null, parity_failed,
    exit code 4."""
    line = run_bench("--no-cpu-baseline")
    pred = line["model"]["scaling_prediction"]
    assert set(pred) == {"1", "2", "4", "8"}
    assert abs(pred["1"]["T_star_ms"] - line["model"]["T_star_ms"]) <= 1e-9 * line["model"]["T_star_ms"] and pred["1"]["sent_GB"] == 0
    for N in ("2", "4", "8"):
        assert pred[N]["sent_GB"] > 0 and pred[N]["latency_chain_ms"] >= pred["1"]["latency_chain_ms"] * (1 - 1e-12)
        assert pred[N]["hbm_fullest_rank_GB"] <= pred["1"]["hbm_fullest_rank_GB"]
        assert pred[N]["bound_ms"] >= pred[N]["latency_chain_ms"] * (1 - 1e-12)
    assert line["parity_failed"] is False and line["factor_check_vectors"] == 8
    env = dict(os.environ, PANGULU_BENCH_TEST_BREAK_FACTORS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "shell", "--size", "40", "40", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-profile-pass"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 4, (out.returncode, out.stdout[-500:], out.stderr[-2000:])
    bad = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith('{"metric"')][-1])
    assert bad["value"] is None and bad["parity_failed"] is True and (bad["factor_check"] > 1e-10 or bad["residual"] > 1e-10)


def test_bench_gives_up_a_hanging_transport_inside_its_budget():
    """First contact at N > 1 must not be able to time out without a line: a transport that passes its start-up
    self-test and then never finishes a step (injected: PANGULU_BENCH_TEST_HANG_TRANSPORT) is given up after --worker-timeout by every
    supervisor, the next transport runs, and the line says what happened -- all inside --total-budget."""
    import time

    env = dict(os.environ, PANGULU_BENCH_TEST_HANG_TRANSPORT="ipc")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "shell", "--size", "40", "40", "--steps", "2", "--warmup", "1",
                          "--transport", "auto", "--worker-timeout", "25", "--total-budget", "400", "--cpu-sample-stride", "3", "--no-profile-pass"],
                         capture_output=True, text=True, timeout=600, env=env)
    wall = time.time() - t0
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    line = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith('{"metric"')][-1])
    att = line["config"]["worker_attempts"]
    # rccl: not available on one device (ends by itself) -> ipc: hangs, killed at its cap -> host: runs
    assert [a["transport"] for a in att] == ["rccl", "ipc", "host"], att
    assert att[1]["done"] is False and 20 <= att[1]["s"] <= 50 and att[2]["done"] is True, att
    assert line["config"]["transport"] == "host" and line["value"] > 0 and line["residual"] < 1e-10
    assert wall < 400 and line["bench_wall_s"] <= 400
    assert line["transport_ab"] is None  # (only behind a first attempt that succeeded)
