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


def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", "40", "40", "--cpu-sample-stride", "3",
                          "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["unit"] == "GFLOP/s" and line["dtype"] == "f64" and line["n_gpus"] == 1 and line["steps"] == 2
    assert line["value"] > 0 and line["residual"] < 1e-10
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert "every 3rd task" in line["cpu_baseline"]["sample"] and line["cpu_baseline"]["value"] > 0
    # mirror maintenance is part of the reported kernel time, and the whole-factorisation bound T* is there
    assert {"densify", "sparsify"} & set(line["kernels"]) or line["config"]["nb"] != 256
    assert 0 < line["roofline"]["model_T_star_over_t_gstrf"] < 1
