"""Pin the oracle (CPU restatement of the reference's CPU platform) against everything the reference holds for this
path: its only fixture (examples/Trefethen_20b.mtx) and the known answers its own runs produced
(tests/golden/known_answers.json), plus the reference's two correctness criteria: the factor check
||L(U 1) - A 1|| / ||A 1|| (src/pangulu_numeric.c:1082-1341) and ||Ax-b||/||b|| (examples/example.c:304-364)."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from pangulu_amd import matrices as M

from .helpers import factorize, lu_check, max_rel_diff, oracle_library

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KNOWN = json.load(open(os.path.join(GOLDEN, "known_answers.json")))


def fixture():
    return M.read_mtx(os.path.join(GOLDEN, "Trefethen_20b.mtx"))


def test_fixture_is_the_trefethen_matrix():
    n, cp, ri, va, _ = fixture()
    assert n == KNOWN["trefethen_20b"]["n"] and len(va) == KNOWN["trefethen_20b"]["nnz_expanded"]
    g = M.trefethen()
    assert g[0] == n and (g[1] == cp).all() and (g[2] == ri).all() and (g[3] == va).all()


@pytest.mark.parametrize("nb", [4, 10, 19, 64])
def test_trefethen_known_answers(nb):
    mat = fixture()
    r = factorize(mat, nb, oracle_library("r64"), ordering="identity")
    k = KNOWN["trefethen_20b"]
    # structural numbers are exact and independent of nb
    assert r["info"]["symbolic_nnz"] == k["symbolic_nnz"]
    assert r["info"]["flop"] == k["flop"]
    # the reference printed 2.0e-16 (nb=10) / 1.4e-16 (nb=4): same order of magnitude is all a residual can pin
    assert r["residual"] < 4 * k["residual_nb10_1rank_r64"]
    assert lu_check(mat, r) < 4 * k["factor_check_r64"]
    if nb == 10:
        assert abs(r["residual"] - k["residual_nb10_1rank_r64"]) < 0.05e-16 * 20  # reference prints 2 digits: 2.0e-16


def test_trefethen_r32_residual():
    n, cp, ri, va, _ = fixture()
    mat = (n, cp, ri, va.astype(np.float32), None)
    r = factorize(mat, 10, oracle_library("r32"), vtype="r32", ordering="identity")
    assert r["residual"] < 3 * KNOWN["trefethen_20b"]["residual_r32"]


def test_poisson24_known_answers():
    mat = M.poisson3d(24)
    k = KNOWN["poisson3d_24"]
    assert mat[0] == k["n"] and len(mat[3]) == k["nnz"]
    r = factorize(mat, 64, oracle_library("r64"), ordering="identity", keep_factors=False)
    assert r["info"]["symbolic_nnz"] == k["symbolic_nnz"]
    assert r["info"]["flop"] == k["flop"]
    assert r["info"]["ntask_tstrf"] == k["nb64"]["tstrf_calls"]
    assert r["info"]["ntask_gessm"] == k["nb64"]["gessm_calls"]
    assert r["info"]["ntask_ssssm"] == k["nb64"]["ssssm_calls"]
    assert r["residual"] < 1e-13


def test_flop_formula_matches_per_task_counters():
    """F = sum_k (c_k + 2 c_k^2) from the symbolic pattern equals the sum of the reference's per-task structural
    counters (src/pangulu_kernel_interface.c:4-176) over the whole task list."""
    import ctypes

    from pangulu_amd import _lib

    # replay: dense LU of the symbolic pattern counts the same operations; here we use the oracle's counters on
    # a single-block factorisation, where the whole matrix is one GETRF task
    mat = M.random_pattern(40, 0.1, 3)
    r = factorize(mat, 64, oracle_library("r64"), ordering="identity")
    L = (r["L"] != 0).astype(np.int64)
    ck = np.asarray(L.sum(axis=0)).ravel() - 1
    assert int((ck + 2 * ck * ck).sum()) == r["info"]["flop"]
    assert _lib is not None and ctypes is not None


CASES = [
    ("poisson6_nb8", lambda dt: M.poisson3d(6, dtype=dt), 8, "nd"),
    ("fem27_5_nb32", lambda dt: M.fem27(5, dtype=dt), 32, "nd"),
    ("shell_6x5_nb16", lambda dt: M.shell(6, 5, dtype=dt), 16, "identity"),
    ("random120_nb16", lambda dt: M.random_pattern(120, 0.05, 9, dtype=dt), 16, "identity"),
    ("kkt3_nb16", lambda dt: M.kkt(3, dtype=dt), 16, "nd"),
]


@pytest.mark.parametrize("vtype,dtype,tol", [("r64", np.float64, 1e-13), ("cr64", np.complex128, 1e-13),
                                             ("r32", np.float32, 5e-5), ("cr32", np.complex64, 5e-5)])
@pytest.mark.parametrize("name,gen,nb,ordering", CASES, ids=[c[0] for c in CASES])
def test_factors_reproduce_the_matrix(name, gen, nb, ordering, vtype, dtype, tol):
    """Independent ground truth: L*U must equal the permuted matrix (no pivoting, diagonally dominant input)."""
    mat = gen(dtype)
    r = factorize(mat, nb, oracle_library(vtype), vtype=vtype, ordering=ordering)
    from .helpers import permuted_matrix

    Ap = permuted_matrix(mat, r["perm"])
    err = abs(r["L"] @ r["U"] - Ap).max() / abs(Ap).max()
    assert err < tol, err
    assert r["residual"] < 50 * tol
    # unit lower / upper triangular shapes
    assert sp.triu(r["L"], 1).nnz == 0 and sp.tril(r["U"], -1).nnz == 0


@pytest.mark.parametrize("name,gen,nb,ordering", CASES, ids=[c[0] for c in CASES])
def test_fma_order_oracle_agrees_with_reference_order_oracle(name, gen, nb, ordering):
    """The PG_ORACLE_FMA build restates the GPU's operation order; it must stay within rounding of the plain build."""
    mat = gen(np.float64)
    a = factorize(mat, nb, oracle_library("r64"), ordering=ordering)
    b = factorize(mat, nb, oracle_library("r64", fma=True), ordering=ordering)
    assert max_rel_diff(a["L"], b["L"]) < 1e-13 and max_rel_diff(a["U"], b["U"]) < 1e-13


def test_openblas_backed_ssssm_matches_triple_loop():
    """bench.py's CPU baseline routes the oracle's SSSSM GEMM to OpenBLAS like the reference (...0100000.c:317-327)."""
    import subprocess
    import sys

    code = (
        "import sys, os; sys.path.insert(0, %r)\n"
        "from bench import find_openblas\n"
        "b = find_openblas()\n"
        "assert b, 'scipy bundled openblas not found'\n"
        "os.environ['PANGULU_ORACLE_BLAS'] = b\n"
        "from tests.helpers import factorize, oracle_library\n"
        "from pangulu_amd import matrices as M\n"
        "r = factorize(M.fem27(6), 32, oracle_library('r64'))\n"
        "print(r['residual'])\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    )
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert float(out.stdout.strip().splitlines()[-1]) < 1e-13


def test_operator_vectors_are_what_the_oracle_produces():
    """tests/golden/operator_vectors.json (made by tests/golden/make_operator_vectors.py): the oracle's 0100000 operators, one
    call per task on hand-built slots, reproduce the committed before/after vectors -- sums of every task to 1e-13 relative,
    the full vectors of the first task of each kind to 1e-14 absolute on O(1) data."""
    import ctypes
    import importlib.util
    import json
    import os

    import numpy as np

    from . import slots as S
    from .helpers import oracle_library

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_operator_vectors", os.path.join(here, "make_operator_vectors.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    gold = json.load(open(os.path.join(here, "operator_vectors.json")))
    assert gold["case"] == mk.CASE
    recs = mk.case_records()  # (under the permutation the fixture carries)
    bm = S.BlockMatrix(recs, mk.CASE["nb"], np.float64, None)
    fo = S.declare_platform(ctypes.CDLL(oracle_library(mk.CASE["vtype"])), "0100000")
    trace = mk.replay(mk.oracle_call(fo), bm)
    assert len(trace) == len(gold["tasks"]) and len(bm.blocks) == gold["blocks"]
    full = 0
    for got, want in zip(trace, gold["tasks"]):
        assert (got["kind"], got["dst"], got.get("op1"), got.get("op2")) == (want["kind"], want["dst"], want.get("op1"), want.get("op2"))
        assert abs(got["sum"] - want["sum"]) <= 1e-13 * max(1.0, abs(want["sum"]))
        assert abs(got["sumsq"] - want["sumsq"]) <= 1e-13 * max(1.0, abs(want["sumsq"]))
        if "after" in want:
            full += 1
            for g, w in zip(got["before"], want["before"]):
                assert np.abs(np.array(g) - np.array(w)).max(initial=0.0) <= 1e-14
            for g, w in zip(got["after"], want["after"]):
                assert np.abs(np.array(g) - np.array(w)).max(initial=0.0) <= 1e-14
    assert full == 4
