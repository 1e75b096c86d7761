"""Shared test plumbing: factorise a matrix on a chosen platform through the public C-ABI and return the factors.

`platform` is "hip" (the product path) or the path of an oracle shared object (CPU restatement of the reference's
CPU platform; test infrastructure only -- see oracle/pangulu_oracle.c).
"""
import os

import numpy as np

import pangulu_amd as pa
from pangulu_amd import _lib
from pangulu_amd import matrices as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_library(vtype="r64", fma=False):
    return os.path.join(ROOT, "oracle", "_build", "libpangulu_oracle_%s%s.so" % (vtype, "_fma" if fma else ""))


def library_for(platform, vtype="r64"):
    """The PRODUCT library for "hip"; the checker's build of the host (with the platform loader) routed to the oracle's CPU
    operators otherwise."""
    if platform == "hip":
        lib = _lib.load(vtype)
        lib.pangulu_amd_use_builtin_platform()
        return lib
    lib = _lib.load(vtype, test_hooks=True)
    rc = lib.pangulu_amd_use_platform_library(platform.encode(), _lib.PLATFORM_CPU_NAIVE)
    assert rc == 0, "cannot load %s" % platform
    return lib


def factorize(mat, nb, platform, vtype="r64", ordering=None, solve=True, keep_factors=True, user_perm=None, nthread=4,
              hip_options=None, scaling=False):
    """Runs pangulu_init + gstrf (+ gstrs with b = A*1) and returns info, factors (scipy CSC, permuted ordering),
    the permutation, x and ||Ax-b||/||b||."""
    n, cp, ri, va, coords = mat
    lib = library_for(platform, vtype)
    if platform == "hip":
        pa.hip_stats(lib, reset=True)
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_GETRF_STRICT_ORDER, 0)
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_DENSE_THRESHOLD_PERMILLE, 2)
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_COUNT_FLOPS, 1)
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_SSSSM_GROUP_CHUNK, 8)
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_TRSM_DENSE_PERMILLE, 5)
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_SMALL_LAUNCH_TASKS, 2048)
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_FRONT_STAGES, 2)
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_TILES_STAGES, 2)
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_BACKGROUND_UPDATES, 1)
        for opt, val in (hip_options or {}).items():
            assert lib.pangulu_platform_0201001_set_option(opt, val) == 0
    if ordering is None:
        ordering = "nd"
    h = pa.pangulu_init(n, len(va), cp, ri, va, nb=nb, vtype=vtype, ordering=ordering,
                        coords=coords if ordering == "nd" else None, user_perm=user_perm, nthread=nthread, lib=lib, scaling=scaling)
    out = {"info": h.info()}
    pa.pangulu_gstrf(h)
    out["info"] = h.info()
    if platform == "hip":
        out["hip_stats"] = pa.hip_stats(lib)
    out["perm"] = pa.permutation(h)
    # the reference's numeric check on the factors where they are (device / host): the all-ones vector and seven random +-1 vectors, the worst
    out["factor_check"] = pa.factor_check_vectors(h, 8)
    if keep_factors:
        out["L"], out["U"] = pa.factors_as_scipy(h)
    if solve:
        b = M.rhs_of_ones(n, cp, ri, va)
        x = pa.pangulu_gstrs(h, b)
        out["x"] = x
        out["residual"] = M.relative_residual(n, cp, ri, va, x, b)
    pa.pangulu_finalize(h)
    return out


def lu_check(mat, res):
    """The reference's factor check (src/pangulu_numeric.c:1082-1341): ||L(U 1) - A' 1|| / ||A' 1|| in the permuted ordering."""
    Ap = permuted_matrix(mat, res["perm"])
    ones = np.ones(Ap.shape[0], dtype=mat[3].dtype)
    lhs = res["L"] @ (res["U"] @ ones)
    rhs = Ap @ ones
    return float(np.linalg.norm(lhs - rhs) / np.linalg.norm(rhs))


def permuted_matrix(mat, perm):
    """P [A 0; 0 I] P^T for the (possibly padded) permutation the solver used."""
    import scipy.sparse as sp

    n, cp, ri, va, _ = mat
    A = M.to_scipy(n, cp, ri, va)
    npad = len(perm)
    if npad > n:
        A = sp.block_diag([A, sp.identity(npad - n, dtype=va.dtype)], format="csr")
    p = perm.astype(np.int64)
    return A.tocsr()[p][:, p]


def max_rel_diff(a, b):
    """max |a-b| / max |b| over two sparse matrices with the same shape."""
    d = abs(a - b)
    scale = abs(b).max()
    return float(d.max() / scale) if scale else float(d.max())
