#!/usr/bin/env python3
"""Re-derive the structural known answers with the oracle platform and compare with known_answers.json."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from pangulu_amd import matrices as M  # noqa: E402
from tests.helpers import factorize, oracle_library  # noqa: E402

known = json.load(open(os.path.join(HERE, "known_answers.json")))
mat = M.read_mtx(os.path.join(HERE, "Trefethen_20b.mtx"))
r = factorize(mat, 10, oracle_library("r64"), ordering="identity")
print("trefethen_20b", r["info"]["symbolic_nnz"], r["info"]["flop"], r["residual"])
assert r["info"]["symbolic_nnz"] == known["trefethen_20b"]["symbolic_nnz"]
assert r["info"]["flop"] == known["trefethen_20b"]["flop"]
mat = M.poisson3d(24)
r = factorize(mat, 64, oracle_library("r64"), ordering="identity", keep_factors=False)
print("poisson3d_24", r["info"]["symbolic_nnz"], r["info"]["flop"], r["info"]["ntask_ssssm"])
assert r["info"]["symbolic_nnz"] == known["poisson3d_24"]["symbolic_nnz"]
assert r["info"]["flop"] == known["poisson3d_24"]["flop"]
