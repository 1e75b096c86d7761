"""GPU parity: the HIP back-end, driven through the public C-ABI, against the oracle on the same inputs.

Tolerances (stated by BASELINE.json / SURVEY.md §8c): factors within 1e-12 (R64/CR64) or 1e-5 (R32/CR32) of the CPU
path relative to the largest factor entry; ||Ax-b||/||b|| within 1e-10 of the CPU reference's.
GETRF/TSTRF/GESSM apply their updates in the same order as the CPU merges, so for matrices whose SSSSM updates are
single-term (block tridiagonal) the factors must match the FMA-rounding oracle bit for bit.
"""
import numpy as np
import pytest

from pangulu_amd import matrices as M

from .helpers import factorize, lu_check, max_rel_diff, oracle_library

pytestmark = pytest.mark.gpu

TOL = {"r64": 1e-12, "cr64": 1e-12, "r32": 1e-5, "cr32": 1e-5}
DTYPES = {"r64": np.float64, "r32": np.float32, "cr64": np.complex128, "cr32": np.complex64}

CASES = [
    ("trefethen20b_nb10", lambda dt: M.trefethen(dtype=dt), 10, "identity"),
    ("trefethen20b_nb4", lambda dt: M.trefethen(dtype=dt), 4, "identity"),
    ("poisson8_nb16_nd", lambda dt: M.poisson3d(8, dtype=dt), 16, "nd"),
    ("poisson8_nb32_identity", lambda dt: M.poisson3d(8, dtype=dt), 32, "identity"),
    ("fem27_8x8x6_nb64", lambda dt: M.fem27(8, 8, 6, dtype=dt), 64, "nd"),
    ("fem27_10_nb256", lambda dt: M.fem27(10, dtype=dt), 256, "nd"),
    ("shell_12x10_nb128", lambda dt: M.shell(12, 10, dtype=dt), 128, "nd"),
    ("random200_nb32", lambda dt: M.random_pattern(200, 0.03, 7, dtype=dt), 32, "identity"),
    ("ragged_n300_nb128", lambda dt: M.random_pattern(300, 0.02, 11, dtype=dt), 128, "nd"),  # n not a multiple of nb
    ("single_block_n50_nb64", lambda dt: M.random_pattern(50, 0.2, 3, dtype=dt), 64, "identity"),  # one padded block
]


@pytest.mark.parametrize("name,gen,nb,ordering", CASES, ids=[c[0] for c in CASES])
def test_factors_match_oracle_r64(name, gen, nb, ordering):
    mat = gen(np.float64)
    gpu = factorize(mat, nb, "hip", ordering=ordering)
    ref = factorize(mat, nb, oracle_library("r64"), ordering=ordering)
    assert gpu["info"]["flop"] == ref["info"]["flop"]
    assert gpu["info"]["symbolic_nnz"] == ref["info"]["symbolic_nnz"]
    assert (gpu["perm"] == ref["perm"]).all()
    for f in ("L", "U"):
        assert gpu[f].nnz == ref[f].nnz
        assert max_rel_diff(gpu[f], ref[f]) <= TOL["r64"], (name, f)
    assert abs(gpu["residual"] - ref["residual"]) <= 1e-10
    assert gpu["residual"] <= 1e-12
    assert lu_check(mat, gpu) <= 1e-12
    # the device counted exactly the structural flops of the closed form
    counted = sum(v["flops"] for v in gpu["hip_stats"].values())
    assert counted == gpu["info"]["flop"], (counted, gpu["info"]["flop"])


def test_dense_blocks_take_the_mfma_path():
    # a dense matrix: every 128x128 block is full, so every off-diagonal SSSSM is a plain GEMM on the value arrays
    rng = np.random.default_rng(5)
    n = 512
    A = rng.uniform(-1, 1, (n, n))
    A += np.diag(np.abs(A).sum(axis=1) + 1.0)
    import scipy.sparse as sp

    S = sp.csc_matrix(A)
    mat = (n, S.indptr.astype(np.uint64), S.indices.astype(np.uint32), S.data.copy(), None)
    gpu = factorize(mat, 128, "hip", ordering="identity")
    ref = factorize(mat, 128, oracle_library("r64"), ordering="identity")
    st = gpu["hip_stats"]
    assert st["ssssm_dense_mfma"]["tasks"] > 0, st
    for f in ("L", "U"):
        assert max_rel_diff(gpu[f], ref[f]) <= 1e-12
    # against LAPACK-free ground truth: L U = A
    LU = (gpu["L"] @ gpu["U"]).toarray()
    assert np.abs(LU - A).max() <= 1e-11 * np.abs(A).max()


@pytest.mark.parametrize("stages", [0, 1, 2, 3])
@pytest.mark.parametrize("n,nb", [(2560, 256), (1536, 128)])
def test_dense_front_kernel(n, nb, stages):
    """A dense matrix: every tile of every update is a dense-front product and runs on the LDS-DMA kernel (pg_hip_front.h)
    with 2, 3 or 4 stages, inside the general launch on its no-step-list path (1, the default), or, with the switch at 0, like
    any partly filled tile; queues of up to nine updates per destination.  Factors against the oracle and against L U = A."""
    import scipy.sparse as sp

    from pangulu_amd import _lib

    rng = np.random.default_rng(nb + stages)
    A = rng.uniform(-1, 1, (n, n))
    A += np.diag(np.abs(A).sum(axis=1) + 1.0)
    S = sp.csc_matrix(A)
    mat = (n, S.indptr.astype(np.uint64), S.indices.astype(np.uint32), S.data.copy(), None)
    # (the dense-front kernel gets a launch of its own from PANGULU_HIP_FRONT_MIN_WGS qualifying workgroups on; the
    #  conftest sets it to 64 for the GPU suite so that the test matrices reach both paths)
    gpu = factorize(mat, nb, "hip", ordering="identity", hip_options={_lib.HIP_OPT_FRONT_STAGES: stages})
    ref = factorize(mat, nb, oracle_library("r64"), ordering="identity")
    st = gpu["hip_stats"]["ssssm_dense_mfma"]
    assert st["tasks"] > 0, st
    if stages:
        # (launches with a handful of updates are cut four ways along K and stay on the general kernel)
        assert st["front_workgroups"] > 0, st
    else:
        assert st["front_workgroups"] == 0 and st["general_workgroups"] > 0, st
    for f in ("L", "U"):
        assert max_rel_diff(gpu[f], ref[f]) <= 1e-12
    LU = (gpu["L"] @ gpu["U"]).toarray()
    assert np.abs(LU - A).max() <= 1e-11 * np.abs(A).max()
    assert gpu["factor_check"] <= 1e-12 and gpu["residual"] <= 1e-12
    counted = sum(v["flops"] for v in gpu["hip_stats"].values())
    assert counted == gpu["info"]["flop"]
    assert gpu["hip_stats"]["ssssm_dense_mfma"]["mfma_flops_executed"] == 2.0 * nb ** 3 * st["tasks"]


@pytest.mark.parametrize("tiles_stages", [0, 2])
@pytest.mark.parametrize("name,gen,nb", [("fem27_20_nb128", lambda: M.fem27(20), 128), ("fem27_24_nb256", lambda: M.fem27(24), 256),
                                         ("shell_40_nb256", lambda: M.shell(40, 40), 256)])
def test_general_update_kernels(name, gen, nb, tiles_stages):
    """The general MFMA update kernel (2: LDS-DMA pipeline, strided piece ownership) and round 2's (0), with the dense-front
    kernel off so that every tile goes through it: partly filled tiles, queues longer than one window in the lower levels."""
    from pangulu_amd import _lib

    mat = gen()
    gpu = factorize(mat, nb, "hip", hip_options={_lib.HIP_OPT_FRONT_STAGES: 0, _lib.HIP_OPT_TILES_STAGES: tiles_stages})
    ref = factorize(mat, nb, oracle_library("r64"))
    st = gpu["hip_stats"]["ssssm_dense_mfma"]
    assert st["tasks"] > 0 and st["general_workgroups"] > 0 and st["front_workgroups"] == 0, st
    for f in ("L", "U"):
        assert max_rel_diff(gpu[f], ref[f]) <= 1e-12, (name, f)
    assert gpu["factor_check"] <= 1e-12 and gpu["residual"] <= 1e-12
    counted = sum(v["flops"] for v in gpu["hip_stats"].values())
    assert counted == gpu["info"]["flop"]


DENSE_MODE_CASES = [
    ("fem27_10_nb256", lambda: M.fem27(10), 256, "nd"),
    ("shell_14x12_nb128", lambda: M.shell(14, 12), 128, "nd"),
    ("poisson10_nb128_identity", lambda: M.poisson3d(10), 128, "identity"),
    ("ragged_n700_nb128", lambda: M.random_pattern(700, 0.02, 4), 128, "nd"),
]


@pytest.mark.parametrize("permille", [0, 40, 100, 1001])
@pytest.mark.parametrize("name,gen,nb,ordering", DENSE_MODE_CASES, ids=[c[0] for c in DENSE_MODE_CASES])
def test_dense_mode_thresholds(name, gen, nb, ordering, permille):
    """Every mix of dense-mode (mirrored, MFMA) and sparse blocks must give the same factors: threshold 0
    mirrors every block (zero-filled dense images of very sparse blocks included), 1001 disables dense mode."""
    from pangulu_amd import _lib

    mat = gen()
    # the solves follow: every TSTRF/GESSM on the dense MFMA path at 0, none at 1001
    gpu = factorize(mat, nb, "hip", ordering=ordering, hip_options={_lib.HIP_OPT_DENSE_THRESHOLD_PERMILLE: permille,
                                                                     _lib.HIP_OPT_TRSM_DENSE_PERMILLE: permille})
    ref = factorize(mat, nb, oracle_library("r64"), ordering=ordering)
    st = gpu["hip_stats"]
    if permille == 0:
        assert st["ssssm_dense_mfma"]["tasks"] > 0 and st["ssssm_sparse"]["tasks"] == 0, st
    if permille == 0:
        assert st["tstrf"]["dense_path_tasks"] == st["tstrf"]["tasks"] + st["gessm"]["tasks"] > 0, st
    if permille == 1001:
        assert st["ssssm_dense_mfma"]["tasks"] == 0 and st["tstrf"]["dense_path_tasks"] == 0
    for f in ("L", "U"):
        assert max_rel_diff(gpu[f], ref[f]) <= 1e-12, (name, f, permille)
    assert gpu["residual"] <= 1e-12 and lu_check(mat, gpu) <= 1e-12
    counted = sum(v["flops"] for v in gpu["hip_stats"].values())
    assert counted == gpu["info"]["flop"], (counted, gpu["info"]["flop"])


BITEXACT_CASES = [
    ("banded", lambda: M.poisson3d(64, 2, 2), 16, "identity"),
    ("fem27_7_nb48", lambda: M.fem27(7), 48, "nd"),
    ("shell_9x8_nb40", lambda: M.shell(9, 8), 40, "nd"),
    ("random150_nb24", lambda: M.random_pattern(150, 0.04, 21), 24, "identity"),
]


@pytest.mark.parametrize("name,gen,nb,ordering", BITEXACT_CASES, ids=[c[0] for c in BITEXACT_CASES])
def test_sparse_path_is_bit_exact_against_fma_oracle(name, gen, nb, ordering):
    # Every sparse kernel applies the updates of an entry one by one in ascending pivot order, each as one fused
    # multiply-add.  The FMA build of the oracle restates exactly that order (oracle/pangulu_oracle.c,
    # PG_ORACLE_FMA), the scheduler is deterministic on one rank, so the whole factorisation must agree bit for
    # bit as long as no block is full enough for the MFMA kernel (nb is not a multiple of 128 here).
    from pangulu_amd import _lib

    mat = gen()
    # strict-order GETRF: the LDS-blocked MFMA variant also applies updates in ascending pivot order, but sums four
    # pivots per matrix-core instruction, whose internal rounding is not specified
    # (and no queue splitting: chunks of a split queue merge with atomics, in arrival order)
    gpu = factorize(mat, nb, "hip", ordering=ordering,
                    hip_options={_lib.HIP_OPT_GETRF_STRICT_ORDER: 1, _lib.HIP_OPT_SSSSM_GROUP_CHUNK: 0})
    ref = factorize(mat, nb, oracle_library("r64", fma=True), ordering=ordering)
    assert gpu["hip_stats"]["ssssm_dense_mfma"]["tasks"] == 0
    for f in ("L", "U"):
        a, b = gpu[f].tocsc(), ref[f].tocsc()
        a.sort_indices()
        b.sort_indices()
        assert (a.indices == b.indices).all() and (a.indptr == b.indptr).all()
        assert (a.data == b.data).all(), f


@pytest.mark.parametrize("vtype", ["r32", "cr64", "cr32"])
def test_other_value_types(vtype):
    dt = DTYPES[vtype]
    shift = 0.5j if np.issubdtype(dt, np.complexfloating) else 0.0
    mat = M.poisson3d(7, dtype=dt, shift=shift)
    gpu = factorize(mat, 32, "hip", vtype=vtype)
    ref = factorize(mat, 32, oracle_library(vtype), vtype=vtype)
    for f in ("L", "U"):
        assert max_rel_diff(gpu[f], ref[f]) <= TOL[vtype], f
    assert gpu["residual"] <= (1e-12 if vtype == "cr64" else 2e-5)


def test_reference_style_per_task_calls_with_host_mirror():
    """Drive the back-end the way the reference host does: eager host mirror on (values copied back after every
    panel task, ...0201000.cu:639-640,680,714) -- the factors must be on the host without an explicit download."""
    mat = M.fem27(6, 6, 5)
    import pangulu_amd as pa
    from pangulu_amd import _lib

    lib = _lib.load("r64")
    lib.pangulu_amd_use_builtin_platform()
    n, cp, ri, va, coords = mat
    h = pa.pangulu_init(n, len(va), cp, ri, va, nb=32, coords=coords, eager_host_mirror=True)
    pa.pangulu_gstrf(h)
    L, U = pa.factors_as_scipy(h)
    pa.pangulu_finalize(h)
    ref = factorize(mat, 32, oracle_library("r64"))
    assert max_rel_diff(L, ref["L"]) <= 1e-12 and max_rel_diff(U, ref["U"]) <= 1e-12


def test_chunked_device_arena(monkeypatch):
    """The device arena is a list of separately allocated chunks (records never straddle one; pg_preprocess.cpp).  With
    1 MiB chunks a small matrix already spans many of them: upload, kernels, snapshot-free download and the solve must
    not notice."""
    monkeypatch.setenv("PANGULU_AMD_ARENA_CHUNK_MB", "1")
    mat = M.fem27(10)
    gpu = factorize(mat, 256, "hip", ordering="nd")
    ref = factorize(mat, 256, oracle_library("r64"), ordering="nd")
    for f in ("L", "U"):
        assert max_rel_diff(gpu[f], ref[f]) <= 1e-12
    assert gpu["residual"] <= 1e-12 and lu_check(mat, gpu) <= 1e-12
