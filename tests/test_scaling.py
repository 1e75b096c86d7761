"""Maximum-product matching + scaling (pg_scaling.cpp; the job of the reference's MC64 port, src/pangulu_reordering.c:149-681,
driver :1130-1272): properties of the matching, and the KKT class it exists for."""
import ctypes
import os

import numpy as np
import pytest
import scipy.sparse as sp

import pangulu_amd as pa
from pangulu_amd import _lib
from pangulu_amd import matrices as M

from .helpers import library_for, oracle_library


def matching(mat, vtype="r64"):
    n, cp, ri, va, _ = mat
    lib = library_for(oracle_library(vtype), vtype)
    lib.pangulu_amd_test_matching.argtypes = [ctypes.c_uint] + [ctypes.c_void_p] * 6
    q = np.zeros(n, np.uint32)
    dr, dc = np.zeros(n), np.zeros(n)
    cp = np.ascontiguousarray(cp, np.uint64)
    ri = np.ascontiguousarray(ri, np.uint32)
    rc = lib.pangulu_amd_test_matching(n, cp.ctypes.data, ri.ctypes.data, va.ctypes.data, q.ctypes.data, dr.ctypes.data, dc.ctypes.data)
    return rc, q, dr, dc


def unsymmetric(n, density, seed, dtype=np.float64):
    rng = np.random.default_rng(seed)
    A = sp.random(n, n, density=density, random_state=rng, format="csc", data_rvs=lambda k: rng.lognormal(0, 3, k) * rng.choice([-1, 1], k))
    # a hidden permutation with large entries guarantees a perfect matching
    p = rng.permutation(n)
    A = A + sp.csc_matrix((rng.lognormal(2, 1, n), (np.arange(n), p)), shape=(n, n))
    if np.issubdtype(dtype, np.complexfloating):
        A = A.astype(dtype) * np.exp(1j * 0.7)
    A = sp.csc_matrix(A, dtype=dtype)
    A.sort_indices()
    return (n, A.indptr.astype(np.uint64), A.indices.astype(np.uint32), A.data.astype(dtype), None)


@pytest.mark.parametrize("n,density,seed,vtype", [(60, 0.1, 0, "r64"), (300, 0.02, 1, "r64"), (500, 0.01, 2, "cr64"), (1200, 0.004, 3, "r64")])
def test_matching_puts_ones_on_the_diagonal_and_bounds_everything_else(n, density, seed, vtype):
    mat = unsymmetric(n, density, seed, _lib.VALUE_TYPES[vtype][0])
    rc, q, dr, dc = matching(mat, vtype)
    assert rc == 0
    assert sorted(q.tolist()) == list(range(n))  # a permutation
    A = M.to_scipy(*mat[:4])
    S = sp.diags(dr) @ A @ sp.diags(dc)
    A1 = abs(S.tocsc()[:, q.astype(np.int64)])
    assert np.abs(A1.diagonal() - 1.0).max() < 1e-10  # matched entries have modulus 1
    assert A1.max() <= 1.0 + 1e-10                     # nothing is larger
    # optimality certificate by brute force on the small case: no permutation has a larger product of moduli
    if n <= 60:
        from scipy.optimize import linear_sum_assignment

        C = np.full((n, n), 1e30)
        Ad = abs(A).toarray()
        C[Ad > 0] = -np.log(Ad[Ad > 0])
        r, c = linear_sum_assignment(C)
        best = -C[r, c].sum()
        mine = np.log(Ad[np.arange(n), q]).sum()
        assert abs(best - mine) <= 1e-8 * max(1.0, abs(best))


def test_matching_keeps_a_dominant_diagonal():
    mat = M.fem27(6)
    rc, q, dr, dc = matching(mat)
    assert rc == 0 and (q == np.arange(mat[0])).all()


def test_structurally_singular_input_is_reported():
    n = 20
    A = sp.lil_matrix((n, n))
    for i in range(n):
        A[i, i] = 1.0
    A[5, 5] = 0.0
    A[5, 6] = 1.0
    A[6, 6] = 0.0  # columns 5.. : row 6 has no entry at all
    A = A.tocsc()
    A.eliminate_zeros()
    mat = (n, A.indptr.astype(np.uint64), A.indices.astype(np.uint32), A.data.copy(), None)
    rc, _, _, _ = matching(mat)
    assert rc != 0


def saddle(nx, delta=0.0):
    """[[H, J^T], [J, -delta I]] with a ZERO (2,2) block at delta = 0: no pivoting order along the diagonal exists."""
    n1, cp, ri, va, _ = M.poisson3d(nx, shift=1.0)
    H = M.to_scipy(n1, cp, ri, va)
    J = (sp.identity(n1, format="csc") * 2.0 - M._stencil(nx, nx, nx, [(1, 0, 0)]))
    A = sp.bmat([[H, J.T], [J, (-delta) * sp.identity(n1)]], format="csc")
    A.eliminate_zeros()
    A.sort_indices()
    return (2 * n1, A.indptr.astype(np.uint64), A.indices.astype(np.uint32), A.data.astype(np.float64), None)


@pytest.mark.parametrize("nb", [16, 64])
def test_saddle_point_system_needs_and_gets_the_matching(nb):
    mat = saddle(5)
    n, cp, ri, va, _ = mat
    b = M.rhs_of_ones(n, cp, ri, va)
    lib = library_for(oracle_library("r64"))
    res = {}
    for scaling, zero_diagonal in ((False, "0"), (False, None), (True, None)):
        if zero_diagonal is not None:
            os.environ["PANGULU_AMD_ZERO_DIAGONAL"] = zero_diagonal
        try:
            h = pa.pangulu_init(n, len(va), cp, ri, va, nb=nb, lib=lib, scaling=scaling)
            inserted = int(h.info()["inserted_diagonals"])
            pa.pangulu_gstrf(h)
            x = pa.pangulu_gstrs(h, b)
            pa.pangulu_finalize(h)
        finally:
            os.environ.pop("PANGULU_AMD_ZERO_DIAGONAL", None)
        with np.errstate(all="ignore"):
            res[(scaling, zero_diagonal)] = (M.relative_residual(n, cp, ri, va, x, b), inserted)
    # with the matching the permuted diagonal is full: nothing is inserted, the solve is exact to round-off
    assert res[(True, None)][0] < 1e-10 and res[(True, None)][1] == 0, res
    # without it and without the reference's zero-diagonal rule: zero pivots clamped to 1e-16 -- garbage
    assert not (res[(False, "0")][0] < 1e-6) and res[(False, "0")][1] == 0, res
    # without it but with the rule (src/pangulu_reordering.c:715-796): every constraint row gets a 1e-8 diagonal entry -- the solution of
    # a system perturbed by 1e-8, which is what the reference delivers for such files when built without its MC64 port
    assert res[(False, None)][1] == n // 2 and 1e-12 < res[(False, None)][0] < 1e-6, res


def test_scaling_leaves_well_posed_systems_solvable_and_is_one_shot():
    mat = M.shell(8, 7)
    n, cp, ri, va, co = mat
    b = M.rhs_of_ones(n, cp, ri, va)
    lib = library_for(oracle_library("r64"))
    h = pa.pangulu_init(n, len(va), cp, ri, va, nb=32, lib=lib, coords=co, scaling=True)
    pa.pangulu_gstrf(h)
    x = pa.pangulu_gstrs(h, b)
    pa.pangulu_finalize(h)
    assert M.relative_residual(n, cp, ri, va, x, b) < 1e-12


def _without_some_diagonal_entries(mat, every):
    """The matrix with the STORED diagonal entry of every `every`-th column removed, and the same matrix with 1e-8 stored there."""
    n, cp, ri, va, co = mat
    A = M.to_scipy(n, cp, ri, va).tolil()
    B = A.copy()
    cols = list(range(1, n, every))
    for j in cols:
        A[j, j] = 0.0
        B[j, j] = 1e-8
    A = A.tocsc()
    A.eliminate_zeros()
    A.sort_indices()
    B = B.tocsc()
    B.sort_indices()
    pack = lambda X: (n, X.indptr.astype(np.uint64), X.indices.astype(np.uint32), X.data.astype(np.float64), co)
    return pack(A), pack(B), len(cols)


def test_zero_diagonal_rule_of_the_reference():
    """src/pangulu_reordering.c:715-796 (pangulu_add_diagonal_element_csc, called on the reference's METIS path, :959): a column
    without a stored diagonal entry gets one with the value 1e-8.  Here on the nested-dissection path: the factors of a matrix with
    missing diagonal entries are BIT FOR BIT those of the same matrix with 1e-8 stored there (the diagonal is not part of the
    graph: same ordering), the handle reports how many were inserted; PANGULU_AMD_ZERO_DIAGONAL=0 and the identity ordering
    (the reference's build without METIS) insert nothing."""
    missing, explicit, k = _without_some_diagonal_entries(M.fem27(6), 7)
    lib = library_for(oracle_library("r64"))
    out = {}
    for name, mat in (("missing", missing), ("explicit", explicit)):
        n, cp, ri, va, co = mat
        h = pa.pangulu_init(n, len(va), cp, ri, va, nb=32, coords=co, lib=lib)
        inserted = int(h.info()["inserted_diagonals"])
        pa.pangulu_gstrf(h)
        L, U = pa.factors_as_scipy(h)
        out[name] = (inserted, L, U, pa.permutation(h))
        pa.pangulu_finalize(h)
    assert out["missing"][0] == k and out["explicit"][0] == 0
    assert (out["missing"][3] == out["explicit"][3]).all()
    assert abs(out["missing"][1] - out["explicit"][1]).max() == 0 and abs(out["missing"][2] - out["explicit"][2]).max() == 0
    n, cp, ri, va, co = missing
    h = pa.pangulu_init(n, len(va), cp, ri, va, nb=32, ordering="identity", lib=lib)
    assert int(h.info()["inserted_diagonals"]) == 0
    pa.pangulu_finalize(h)
    os.environ["PANGULU_AMD_ZERO_DIAGONAL"] = "0"
    try:
        h = pa.pangulu_init(n, len(va), cp, ri, va, nb=32, coords=co, lib=lib)
        assert int(h.info()["inserted_diagonals"]) == 0
        pa.pangulu_finalize(h)
    finally:
        del os.environ["PANGULU_AMD_ZERO_DIAGONAL"]
