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


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: fresh rank processes, a complete line.  On a one-GPU box both
    ranks share the device: RCCL refuses that, the transports fall back together and the line says so."""
    line = run_bench("--gpus", "2", "--cpu-sample-stride", "3")
    for key in CONTRACT_KEYS:
        assert key in line, key
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["residual"] < 1e-10 and line["factor_check"] < 1e-12
    cfg = line["config"]
    assert cfg["transport"] in ("host", "ipc", "rccl") and len(cfg["transport_tried"]) >= 1
    assert cfg["rccl_nranks"] == (2 if cfg["transport"] == "rccl" else 0)
    cpu = line["cpu_baseline"]
    assert cpu["cores"] == 2 and cpu["value"] > 0, cpu                      # R ranks x 1 thread (examples/example.c:284)
    assert cpu["one_rank_x_one_thread"]["cores"] == 1 and cpu["one_rank_x_one_thread"]["value"] > 0
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
