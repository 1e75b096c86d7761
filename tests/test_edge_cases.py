"""Edge cases of the input side (what the reference's checks and loops imply: src/pangulu.c:28-70 argument checks,
src/pangulu_symbolic.c on A + A^T, padding of the last block row): tiny and ragged orders, a diagonal matrix (GETRF tasks
only), unsymmetric patterns, explicit zeros, n a multiple of nb and one off.  CPU: the host + oracle; GPU: the HIP path."""
import numpy as np
import pytest
import scipy.sparse as sp

from pangulu_amd import matrices as M

from .helpers import factorize, lu_check, max_rel_diff, oracle_library


def _csc(A):
    A = sp.csc_matrix(A)
    A.sort_indices()
    return (A.shape[0], A.indptr.astype(np.uint64), A.indices.astype(np.uint32), A.data.astype(np.float64), None)


def diagonal(n):
    return _csc(sp.diags(np.arange(1, n + 1, dtype=np.float64)))


def unsymmetric_pattern(n, seed):
    rng = np.random.default_rng(seed)
    A = sp.random(n, n, density=0.03, random_state=rng, format="csr", data_rvs=lambda k: rng.uniform(-1, 1, k))
    A = sp.triu(A, 1) * 1.0 + sp.tril(sp.random(n, n, density=0.01, random_state=rng, format="csr"), -1)
    A = A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + np.asarray(abs(A).sum(axis=0)).ravel() + 1.0)
    return _csc(A)


def explicit_zeros(n):
    n0, cp, ri, va, _ = M.poisson3d(4)
    va = va.copy()
    va[va < 0] = 0.0  # every off-diagonal entry stays in the pattern with value 0
    return (n0, cp, ri, va, None)


CASES = [
    ("one_by_one", lambda: _csc(sp.csc_matrix(np.array([[3.0]]))), 16, "identity"),
    ("n5_nb16", lambda: M.random_pattern(5, 0.5, 1), 16, "identity"),
    ("n_equals_nb", lambda: M.random_pattern(64, 0.1, 2), 64, "identity"),
    ("n_one_more_than_nb", lambda: M.random_pattern(65, 0.1, 3), 64, "identity"),
    ("n_one_less_than_2nb", lambda: M.random_pattern(127, 0.05, 4), 64, "nd"),
    ("diagonal_200", lambda: diagonal(200), 32, "identity"),
    ("unsymmetric_pattern", lambda: unsymmetric_pattern(300, 5), 48, "nd"),
    ("explicit_zero_values", lambda: explicit_zeros(64), 16, "nd"),
    ("diagonal_nb128", lambda: diagonal(300), 128, "nd"),
]


@pytest.mark.parametrize("name,gen,nb,ordering", CASES, ids=[c[0] for c in CASES])
def test_edge_cases_on_the_host_with_the_oracle(name, gen, nb, ordering):
    mat = gen()
    r = factorize(mat, nb, oracle_library("r64"), ordering=ordering)
    assert r["residual"] <= 1e-13 and lu_check(mat, r) <= 1e-13
    if name.startswith("diagonal"):
        assert r["info"]["flop"] == 0 and r["info"]["ntask_ssssm"] == 0 and r["info"]["ntask_tstrf"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name,gen,nb,ordering", CASES, ids=[c[0] for c in CASES])
def test_edge_cases_on_the_hip_path(name, gen, nb, ordering):
    mat = gen()
    gpu = factorize(mat, nb, "hip", ordering=ordering)
    ref = factorize(mat, nb, oracle_library("r64"), ordering=ordering)
    assert (gpu["perm"] == ref["perm"]).all() and gpu["info"]["flop"] == ref["info"]["flop"]
    for f in ("L", "U"):
        assert gpu[f].nnz == ref[f].nnz and max_rel_diff(gpu[f], ref[f]) <= 1e-12
    assert gpu["residual"] <= 1e-13
