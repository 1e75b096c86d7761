"""Separators in k-d order (pg_analysis.cpp): the permutation changes inside separators only -- same fill, same
structural flops, same block pattern size class -- and the factor blocks' 16 x 16 pieces get fuller.  The switch is read once per
process, so each setting runs in a child process (CPU, the oracle's operators behind the native host)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, sys
import numpy as np
import pangulu_amd as pa
from pangulu_amd import matrices as M
from tests.helpers import library_for, oracle_library, factorize
lib = library_for(oracle_library("r64"))
mat = M.fem27(16)
n, cp, ri, va, co = mat
h = pa.pangulu_init(n, len(va), cp, ri, va, nb=64, ordering="nd", coords=co, lib=lib, nthread=4)
info = h.info()
pieces = 0
nnz = 0
for brow, bcol, up, bcp, bri, bva in pa.owned_blocks(h):
    if brow == bcol:
        continue
    cols = np.repeat(np.arange(64), np.diff(bcp.astype(np.int64)))
    m = np.zeros((4, 4), bool)
    m[bri.astype(np.int64) >> 4, cols >> 4] = True
    pieces += int(m.sum())
    nnz += len(bri)
perm = pa.permutation(h)
pa.pangulu_finalize(h)
res = factorize(mat, 64, oracle_library("r64"))
print(json.dumps({"symbolic_nnz": int(info["symbolic_nnz"]), "flop": float(info["flop"]), "pieces": pieces, "offdiag_nnz": nnz,
                  "perm_ok": bool(sorted(perm.tolist()) == list(range(len(perm)))), "residual": float(res["residual"])}))
"""


def run(order):
    env = dict(os.environ)
    env["PANGULU_AMD_SEPARATOR_ORDER"] = order
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    out = subprocess.run([sys.executable, "-c", WORKER], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


def test_kd_order_keeps_fill_and_flops_and_fills_the_pieces():
    nat, kd = run("natural"), run("kd")
    assert nat["perm_ok"] and kd["perm_ok"]
    assert kd["symbolic_nnz"] == nat["symbolic_nnz"] and kd["flop"] == nat["flop"]  # same elimination, other labels inside separators
    assert kd["offdiag_nnz"] > 0 and nat["offdiag_nnz"] > 0
    # fewer live 16 x 16 pieces for (nearly) the same entries: fuller pieces
    assert kd["pieces"] < nat["pieces"], (kd, nat)
    assert kd["offdiag_nnz"] / kd["pieces"] > nat["offdiag_nnz"] / nat["pieces"], (kd, nat)
    assert kd["residual"] < 1e-12 and nat["residual"] < 1e-12
