"""Per-operator C-ABI parity: the 0201001 operators called the way a reference host calls them (one task per call,
`ssssm_batched`, `hybrid_batched` with a DEPENDENT array and the hazard scan on, `spmv/vecadd/sptrsv`) on hand-built
slots, against the oracle's 0100000 operators on the same slots.

Reference call sites: src/pangulu_kernel_interface.c:190-337 (dispatcher), …0100000.c:57-506 (CPU semantics),
…0201000.cu:875-909 (serial meaning of the batched entry points).
Tolerance: 1e-12 relative to the largest entry of the block set (R64/CR64), 1e-5 (R32/CR32).
"""
import ctypes

import numpy as np
import pytest

from pangulu_amd import _lib
from pangulu_amd import matrices as M

from . import slots as S
from .helpers import oracle_library

pytestmark = pytest.mark.gpu

TOL = {"r64": 1e-12, "cr64": 1e-12, "r32": 1e-5, "cr32": 1e-5}


def _platforms(vtype):
    hip = _lib.load(vtype)
    hip.pangulu_amd_use_builtin_platform()
    ora = ctypes.CDLL(oracle_library(vtype))
    fh = S.declare_platform(hip, "0201001")
    fo = S.declare_platform(ora, "0100000")
    hip.pangulu_platform_0201001_set_default_device(0)
    for opt, val in ((_lib.HIP_OPT_HOST_MIRROR, 1), (_lib.HIP_OPT_ASSUME_INDEPENDENT, 0), (_lib.HIP_OPT_GETRF_STRICT_ORDER, 0),
                     (_lib.HIP_OPT_DENSE_THRESHOLD_PERMILLE, 10), (_lib.HIP_OPT_TRSM_DENSE_PERMILLE, 10),
                     (_lib.HIP_OPT_SSSSM_GROUP_CHUNK, 8), (_lib.HIP_OPT_RESET_BLOCK_STATE, 0)):
        assert hip.pangulu_platform_0201001_set_option(opt, val) == 0
    return hip, fh, fo


def _compare(bm_gpu, bm_ref, tol, what):
    scale = max(float(np.abs(b.values).max()) for b in bm_ref.blocks.values() if b.nnz)
    worst = 0.0
    for key, rb in bm_ref.blocks.items():
        gb = bm_gpu.blocks[key]
        dev = gb.download_values()
        if rb.nnz:
            worst = max(worst, float(np.abs(dev - rb.values).max()))
    assert worst <= tol * scale, "%s: max |hip - oracle| = %g (scale %g)" % (what, worst, scale)


def _case(name, vtype):
    dt = _lib.VALUE_TYPES[vtype][0]
    shift = 0.5j if np.issubdtype(dt, np.complexfloating) else 0.0
    return {
        "fem27_5_nb32": (lambda: M.fem27(5, dtype=dt), 32),
        "poisson6_nb48": (lambda: M.poisson3d(6, dtype=dt, shift=shift), 48),
        "shell_8x6_nb64": (lambda: M.shell(8, 6, dtype=dt), 64),
        "fem27_8_nb128": (lambda: M.fem27(8, dtype=dt), 128),
        "kkt4_nb32": (lambda: M.kkt(4, dtype=dt), 32),
    }[name]


def _run_serial(f, bm, tasks):
    nb = bm.nb
    for kid, dst, a, b in tasks:
        if kid == S.GETRF:
            f("getrf")(nb, dst.ref(), 0)
        elif kid == S.TSTRF:
            f("tstrf")(nb, dst.ref(), a.ref(), 0)
        elif kid == S.GESSM:
            f("gessm")(nb, dst.ref(), a.ref(), 0)
        else:
            f("ssssm")(nb, dst.ref(), a.ref(), b.ref(), 0)


@pytest.mark.parametrize("vtype", ["r64", "cr64", "r32"])
@pytest.mark.parametrize("name", ["fem27_5_nb32", "poisson6_nb48", "shell_8x6_nb64", "fem27_8_nb128", "kkt4_nb32"])
def test_single_task_operators(name, vtype):
    """getrf / tstrf / gessm / ssssm one call per task in the reference's serial right-looking order; after every panel
    operator the HOST copy of the destination must be current (…0201000.cu:639-640,680,714: MPI send and SpTRSV read it)."""
    gen, nb = _case(name, vtype)
    if vtype != "r64" and nb == 128 and name != "fem27_8_nb128":
        pytest.skip("one nb=128 case per extra type")
    hip, fh, fo = _platforms(vtype)
    dt = _lib.VALUE_TYPES[vtype][0]
    recs = S.exported_records(gen(), nb, vtype)
    g = S.BlockMatrix(recs, nb, dt, hip)
    r = S.BlockMatrix(recs, nb, dt, None)
    try:
        tg, tr = g.tasks(), r.tasks()
        assert len(tg) == len(tr) > 0
        scale = None
        for (kid, dst, a, b), (_, rdst, ra, rb) in zip(tg, tr):
            _run_serial(fh, g, [(kid, dst, a, b)])
            _run_serial(fo, r, [(kid, rdst, ra, rb)])
            if kid != S.SSSSM:
                # host mirror semantics: the destination's host values are the factor now (both halves for GETRF)
                fh("synchronize")()
                halves = [dst, g.blocks[(dst.brow, dst.bcol, 0)]] if kid == S.GETRF else [dst]
                rhalves = [rdst, r.blocks[(dst.brow, dst.bcol, 0)]] if kid == S.GETRF else [rdst]
                for hb, rb_ in zip(halves, rhalves):
                    if rb_.nnz:
                        scale = max(float(np.abs(rb_.values).max()), 1e-300)
                        assert float(np.abs(hb.values - rb_.values).max()) <= TOL[vtype] * max(scale, 1.0), (name, kid, dst.brow, dst.bcol)
        fh("synchronize")()
        _compare(g, r, TOL[vtype], name)
    finally:
        g.free()


@pytest.mark.parametrize("name", ["fem27_5_nb32", "shell_8x6_nb64", "fem27_8_nb128", "kkt4_nb32"])
def test_hybrid_batched_dependent_array_with_hazard_scan(name):
    """The whole factorisation as ONE hybrid_batched array in serial order, ASSUME_INDEPENDENT = 0: the back-end must keep
    the reference's serial meaning (…0201000.cu:875-898) by cutting the array at every dependence."""
    vtype = "r64"
    gen, nb = _case(name, vtype)
    hip, fh, fo = _platforms(vtype)
    recs = S.exported_records(gen(), nb, vtype)
    g = S.BlockMatrix(recs, nb, np.float64, hip)
    r = S.BlockMatrix(recs, nb, np.float64, None)
    try:
        tg, tr = g.tasks(), r.tasks()
        arr_g, arr_r = g.task_array(tg), r.task_array(tr)
        fh("hybrid_batched")(nb, len(tg), arr_g)
        fo("hybrid_batched")(nb, len(tr), arr_r)
        fh("synchronize")()
        _compare(g, r, 1e-12, name)
    finally:
        g.free()


@pytest.mark.parametrize("name", ["fem27_5_nb32", "fem27_8_nb128"])
def test_ssssm_batched_per_level(name):
    """Panel operators one by one, the level's Schur updates through ssssm_batched (src/pangulu_kernel_interface.c:302)."""
    vtype = "r64"
    gen, nb = _case(name, vtype)
    hip, fh, fo = _platforms(vtype)
    recs = S.exported_records(gen(), nb, vtype)
    g = S.BlockMatrix(recs, nb, np.float64, hip)
    r = S.BlockMatrix(recs, nb, np.float64, None)
    try:
        for f, bm in ((fh, g), (fo, r)):
            pending = []
            for t in bm.tasks():
                if t[0] == S.SSSSM:
                    pending.append(t)
                    continue
                if pending:
                    f("ssssm_batched")(nb, len(pending), bm.task_array(pending))
                    pending = []
                _run_serial(f, bm, [t])
            assert not pending
        fh("synchronize")()
        _compare(g, r, 1e-12, name)
    finally:
        g.free()


@pytest.mark.parametrize("vtype", ["r64", "cr64", "r32", "cr32"])
def test_spmv_vecadd_sptrsv_operators(vtype):
    """Solve-side operators on device vectors (semantics …0100000.c:435-506: y -= A x, b += x, unit-lower / upper solve)."""
    dt = _lib.VALUE_TYPES[vtype][0]
    hip, fh, fo = _platforms(vtype)
    nb = 64
    rng = np.random.default_rng(3)
    recs = S.exported_records(M.fem27(6, dtype=dt), nb, vtype)
    g = S.BlockMatrix(recs, nb, dt, hip)
    r = S.BlockMatrix(recs, nb, dt, None)
    sv = np.dtype(dt).itemsize

    def dvec(x):
        p = ctypes.c_void_p()
        fh("malloc")(ctypes.byref(p), len(x) * sv)
        fh("memcpy")(p, ctypes.c_void_p(x.ctypes.data), len(x) * sv, 0)
        return p

    def hvec(p, n):
        out = np.zeros(n, dtype=dt)
        fh("memcpy")(ctypes.c_void_p(out.ctypes.data), p, n * sv, 1)
        return out

    def rnd(n):
        x = rng.uniform(-1, 1, n)
        if np.issubdtype(dt, np.complexfloating):
            x = x + 1j * rng.uniform(-1, 1, n)
        return x.astype(dt)

    tol = TOL[vtype] * 50
    try:
        # factorise first so the diagonal halves hold L and U
        _run_serial(fh, g, g.tasks())
        _run_serial(fo, r, r.tasks())
        fh("synchronize")()
        off = [k for k in g.blocks if k[0] != k[1]][:6]
        for key in off:
            x, y = rnd(nb), rnd(nb)
            dx, dy = dvec(x), dvec(y)
            yr = y.copy()
            fh("spmv")(nb, g.blocks[key].ref(), dx, dy)
            fo("spmv")(nb, r.blocks[key].ref(), ctypes.c_void_p(x.ctypes.data), ctypes.c_void_p(yr.ctypes.data))
            fh("synchronize")()
            got = hvec(dy, nb)
            assert np.abs(got - yr).max() <= tol * max(1.0, np.abs(yr).max()), key
            fh("free")(dx)
            fh("free")(dy)
        b, x = rnd(1000), rnd(1000)
        db, dx = dvec(b), dvec(x)
        br = b.copy()
        fh("vecadd")(1000, db, dx)
        fo("vecadd")(1000, ctypes.c_void_p(br.ctypes.data), ctypes.c_void_p(x.ctypes.data))
        fh("synchronize")()
        assert np.abs(hvec(db, 1000) - br).max() <= tol
        for k in range(min(g.nblk, 4)):
            for uplo, half in ((0, 0), (1, 1)):
                x = rnd(nb)
                dx = dvec(x)
                xr = x.copy()
                fh("sptrsv")(nb, g.blocks[(k, k, half)].ref(), dx, uplo)
                fo("sptrsv")(nb, r.blocks[(k, k, half)].ref(), ctypes.c_void_p(xr.ctypes.data), uplo)
                fh("synchronize")()
                got = hvec(dx, nb)
                assert np.abs(got - xr).max() <= tol * max(1.0, np.abs(xr).max()), (k, uplo)
                fh("free")(dx)
    finally:
        g.free()


def test_operators_reproduce_the_committed_vectors():
    """The HIP operators, one call per task, against tests/golden/operator_vectors.json (the oracle's before/after vectors,
    committed with the script that made them): every task's destination sum / sum of squares and the full vectors of the first
    GETRF, TSTRF, GESSM and SSSSM, 1e-12 relative to the largest entry."""
    import importlib.util
    import json
    import os

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_operator_vectors", os.path.join(here, "make_operator_vectors.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    gold = json.load(open(os.path.join(here, "operator_vectors.json")))
    hip, fh, _ = _platforms("r64")
    recs = mk.case_records()  # (under the permutation the fixture carries)
    g = S.BlockMatrix(recs, mk.CASE["nb"], np.float64, hip)
    try:
        tasks = g.tasks()
        assert len(tasks) == len(gold["tasks"])
        for (kid, dst, a, b), want in zip(tasks, gold["tasks"]):
            assert mk.KIND[kid] == want["kind"] and [int(dst.brow), int(dst.bcol), int(dst.is_upper)] == want["dst"]
            halves = [dst] if kid != S.GETRF else [dst, g.blocks[(dst.brow, dst.bcol, 0 if dst.is_upper else 1)]]
            if "before" in want:
                for h, w in zip(halves, want["before"]):
                    assert np.abs(h.download_values() - np.array(w)).max(initial=0.0) <= 1e-12
            _run_serial(fh, g, [(kid, dst, a, b)])
            fh("synchronize")()
            after = [h.download_values() for h in halves]
            scale = max(1.0, max(float(np.abs(x).max(initial=0.0)) for x in after))
            assert abs(sum(float(x.sum()) for x in after) - want["sum"]) <= 1e-12 * scale * sum(x.size for x in after)
            if "after" in want:
                for x, w in zip(after, want["after"]):
                    assert np.abs(x - np.array(w)).max(initial=0.0) <= 1e-12 * scale
    finally:
        g.free()
