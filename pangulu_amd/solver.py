"""Python mirror of the solver API: same call sequence and argument meaning as include/pangulu.h.

    h = pangulu_init(n, nnz, colptr, rowidx, value, nb=256)     # reorder + symbolic + block records + upload
    pangulu_gstrf(h)                                            # numeric LU on the GPU (the hot path)
    x = pangulu_gstrs(h, b)                                     # triangular solves
    pangulu_finalize(h)

(reference: src/pangulu.c:11-345, driver examples/example.c:282-300).  This is a binding, not an
implementation: every call goes through the C-ABI of libpangulu_amd_<type>.so.
"""
import ctypes

import numpy as np

from . import _lib


class Handle:
    """Opaque solver handle (the reference's ``void *pangulu_handle``) plus the arrays that must outlive it."""

    def __init__(self, lib, vtype):
        self.lib = lib
        self.vtype = vtype
        self.dtype = _lib.VALUE_TYPES[vtype][0]
        self.ptr = ctypes.c_void_p(None)
        self.n = 0
        self._keep = []

    @property
    def ref(self):
        return ctypes.byref(self.ptr)

    def info(self):
        out = _lib.Info()
        self.lib.pangulu_amd_get_info(self.ref, ctypes.byref(out))
        return out.as_dict()


def _as(arr, dtype):
    a = np.ascontiguousarray(arr, dtype=dtype)
    return a


def pangulu_init(n, nnz, csc_colptr, csc_rowidx, csc_value, nb=256, nthread=1, vtype="r64",
                 ordering=None, coords=None, user_perm=None, recv_buffer_level=0.5, eager_host_mirror=False, lib=None, scaling=False):
    """CSC input with 64-bit column pointers and 32-bit row indices, as the reference (src/pangulu_common.h:67-70).

    ordering: None/"nd" (built-in nested dissection, geometric when ``coords`` is given), "identity" (what the
    reference does when built without METIS/MC64) or "user" with ``user_perm`` (perm[new] = old).
    On ranks other than 0 the matrix arguments may be None (rank 0 broadcasts them, as examples/example.c does).
    """
    lib = lib or _lib.load(vtype)  # (tests pass the checker's build of the library, _lib.load(vtype, test_hooks=True))
    h = Handle(lib, vtype)
    dtype, sizeof_value, is_complex = _lib.VALUE_TYPES[vtype]
    if ordering in (None, "nd"):
        lib.pangulu_amd_set_ordering(_lib.ORDER_ND)
    elif ordering == "identity":
        lib.pangulu_amd_set_ordering(_lib.ORDER_IDENTITY)
    elif ordering == "user":
        p = _as(user_perm, np.uint32)
        lib.pangulu_amd_set_user_perm(p.ctypes.data_as(ctypes.c_void_p), len(p))
    else:
        raise ValueError("unknown ordering %r" % (ordering,))
    if coords is not None:
        c = _as(coords, np.float64)
        lib.pangulu_amd_set_coordinates(c.ctypes.data_as(ctypes.c_void_p), c.shape[0], c.shape[1])
    lib.pangulu_amd_set_eager_host_mirror(1 if eager_host_mirror else 0)
    lib.pangulu_amd_set_scaling(1 if scaling else 0)  # maximum-product matching + scaling before the ordering (MC64's job)
    opt = _lib.InitOptions()
    opt.nthread = nthread
    opt.nb = nb
    opt.gpu_kernel_warp_per_block = 4
    opt.gpu_data_move_warp_per_block = 4
    opt.sizeof_value = sizeof_value
    opt.is_complex_matrix = is_complex
    opt.mpi_recv_buffer_level = recv_buffer_level
    if csc_colptr is not None:
        cp = _as(csc_colptr, np.uint64)
        ri = _as(csc_rowidx, np.uint32)
        va = _as(csc_value, dtype)
        h._keep = [cp, ri, va]
        args = (cp.ctypes.data_as(ctypes.c_void_p), ri.ctypes.data_as(ctypes.c_void_p), va.ctypes.data_as(ctypes.c_void_p))
    else:
        args = (None, None, None)
    lib.pangulu_init(n, nnz, args[0], args[1], args[2], ctypes.byref(opt), h.ref)
    h.n = int(h.info()["n"])
    return h


def pangulu_gstrf(h):
    opt = _lib.GstrfOptions()
    h.lib.pangulu_gstrf(ctypes.byref(opt), h.ref)


def pangulu_gstrs(h, rhs):
    """Solves A x = rhs with the factors; returns x (rank 0; other ranks get their input back)."""
    opt = _lib.GstrsOptions()
    x = np.array(rhs, dtype=h.dtype, copy=True) if rhs is not None else np.zeros(h.n, dtype=h.dtype)
    h.lib.pangulu_gstrs(x.ctypes.data_as(ctypes.c_void_p), ctypes.byref(opt), h.ref)
    return x


def pangulu_gssv(h, rhs):
    pangulu_gstrf(h)
    return pangulu_gstrs(h, rhs)


def pangulu_finalize(h):
    if h.ptr:
        h.lib.pangulu_finalize(h.ref)
    h.ptr = ctypes.c_void_p(None)


# ---- introspection used by tests and bench.py ------------------------------------------------------------------

def owned_blocks(h):
    """Yields (brow, bcol, is_upper, colptr, rowidx, values) for every block record this rank owns.

    For is_upper == 1 diagonal halves colptr/rowidx are the CSR row pointer / column index."""
    nb = int(h.info()["nb"])
    cnt = h.lib.pangulu_amd_owned_block_count(h.ref)
    for i in range(cnt):
        brow, bcol = ctypes.c_uint32(), ctypes.c_uint32()
        up, nnz = ctypes.c_int(), ctypes.c_ulonglong()
        cp, ri, va = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        rc = h.lib.pangulu_amd_owned_block(h.ref, i, ctypes.byref(brow), ctypes.byref(bcol), ctypes.byref(up), ctypes.byref(nnz),
                                           ctypes.byref(cp), ctypes.byref(ri), ctypes.byref(va))
        assert rc == 0
        k = int(nnz.value)
        colptr = np.ctypeslib.as_array(ctypes.cast(cp, ctypes.POINTER(ctypes.c_uint32)), shape=(nb + 1,)).copy()
        if k:
            rowidx = np.ctypeslib.as_array(ctypes.cast(ri, ctypes.POINTER(ctypes.c_uint16)), shape=(k,)).copy()
            raw = np.ctypeslib.as_array(ctypes.cast(va, ctypes.POINTER(ctypes.c_uint8)), shape=(k * np.dtype(h.dtype).itemsize,))
            values = raw.view(h.dtype).copy()
        else:
            rowidx = np.zeros(0, np.uint16)
            values = np.zeros(0, h.dtype)
        yield int(brow.value), int(bcol.value), int(up.value), colptr, rowidx, values


def permutation(h):
    p = h.lib.pangulu_amd_get_perm(h.ref)
    return np.ctypeslib.as_array(p, shape=(int(h.info()["n_padded"]),)).copy()


def factors_as_scipy(h):
    """Assemble L (unit lower) and U from this rank's blocks as scipy CSC matrices in the PERMUTED ordering
    (order n_padded: a block-aligned dissection adds isolated unit rows, see pangulu_amd_ext.h)."""
    import scipy.sparse as sp

    info = h.info()
    nb, n = int(info["nb"]), int(info["n_padded"])
    npad = int(info["block_length"]) * nb
    rows_l, cols_l, vals_l, rows_u, cols_u, vals_u = [], [], [], [], [], []
    for brow, bcol, up, cp, ri, va in owned_blocks(h):
        major = np.repeat(np.arange(nb, dtype=np.int64), np.diff(cp.astype(np.int64)))
        minor = ri.astype(np.int64)
        if brow == bcol and up:
            r, c = major + brow * nb, minor + bcol * nb  # CSR
            rows_u.append(r), cols_u.append(c), vals_u.append(va)
        elif brow < bcol:
            r, c = minor + brow * nb, major + bcol * nb
            rows_u.append(r), cols_u.append(c), vals_u.append(va)
        else:
            r, c = minor + brow * nb, major + bcol * nb
            rows_l.append(r), cols_l.append(c), vals_l.append(va)

    def build(rows, cols, vals):
        if not rows:
            return sp.csc_matrix((npad, npad), dtype=h.dtype)
        return sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(npad, npad))

    L = build(rows_l, cols_l, vals_l)[:n, :n] + sp.identity(n, dtype=h.dtype, format="csc")
    U = build(rows_u, cols_u, vals_u)[:n, :n]
    return L.tocsc(), U.tocsc()


def update_values(h, csc_value):
    """New values on the pattern the handle was initialised with (same order as the csc_rowidx given to pangulu_init; rank 0):
    the next pangulu_gstrf factorises the new matrix, re-using ordering, symbolic factorisation, records and -- on one rank --
    the recorded launch schedule."""
    if csc_value is not None:
        va = np.ascontiguousarray(csc_value, dtype=h.dtype)  # (alive until the call returns: the library copies the values)
        ptr = va.ctypes.data_as(ctypes.c_void_p)
    else:
        ptr = None
    rc = h.lib.pangulu_amd_update_values(h.ref, ptr)
    if rc != 0:
        raise RuntimeError("pangulu_amd_update_values failed (%d)" % rc)


def factor_check(h):
    """||L(U 1) - A 1||_2 / ||A 1||_2 on the factors where they are (the reference's pangulu_numeric_check,
    src/pangulu_numeric.c:1082-1341); collective over the ranks."""
    out = ctypes.c_double(0.0)
    rc = h.lib.pangulu_amd_factor_check(h.ref, ctypes.byref(out))
    if rc != 0:
        raise RuntimeError("pangulu_amd_factor_check: the handle has not been factorised")
    return float(out.value)


def factor_check_vectors(h, nvec=8, seed=20251003):
    """The factor check on the all-ones vector and nvec - 1 seeded random +-1 vectors: the largest quotient (collective)."""
    out = ctypes.c_double(0.0)
    rc = h.lib.pangulu_amd_factor_check_vectors(h.ref, int(nvec), int(seed), ctypes.byref(out))
    if rc != 0:
        raise RuntimeError("pangulu_amd_factor_check_vectors: the handle has not been factorised")
    return float(out.value)


MODEL_FIELDS = ("T_star_s", "sum_over_ranks_s", "link_term_s_max", "sent_bytes", "critical_path_s", "latency_chain_s", "rank_flop_share",
                "rank_T_star_share", "hbm_bytes_fullest_rank", "hbm_records_owned", "hbm_records_received", "hbm_dense_mirrors")


def model_for_ranks(h, nranks):
    """The structure-only model of this handle's factorisation on `nranks` ranks (pangulu_amd_model_for_ranks): a dict of
    MODEL_FIELDS, or None when the model is not available."""
    out = (ctypes.c_double * 12)()
    if h.lib.pangulu_amd_model_for_ranks(h.ref, int(nranks), out) != 0:
        return None
    return dict(zip(MODEL_FIELDS, (float(x) for x in out)))


def hip_memory(h_or_lib):
    """Device memory the back-end holds for itself (bytes): mirror pool, descriptor twins of a recorded schedule, GETRF scratch;
    and the number of blocks in dense mode."""
    lib = h_or_lib.lib if isinstance(h_or_lib, Handle) else h_or_lib
    v = (ctypes.c_ulonglong * 4)()
    lib.pangulu_platform_0201001_get_memory(v)
    return {"mirror_pool_bytes": int(v[0]), "schedule_descriptor_bytes": int(v[1]), "getrf_scratch_bytes": int(v[2]), "dense_mode_blocks": int(v[3])}


def hip_stats(h_or_lib, reset=False):
    lib = h_or_lib.lib if isinstance(h_or_lib, Handle) else h_or_lib
    st = _lib.HipStats()
    lib.pangulu_platform_0201001_get_stats(ctypes.byref(st), 1 if reset else 0)
    out = {}
    for k, name in _lib.KERNEL_CLASSES.items():
        out[name] = dict(launches=int(st.launches[k]), tasks=int(st.tasks[k]), alg_bytes=float(st.alg_bytes[k]),
                         flops=float(st.flops[k]), elapsed_ms=float(st.elapsed_ms[k]))
    out["ssssm_dense_mfma"]["mfma_flops_executed"] = float(st.mfma_flops_executed)
    out["tstrf"]["dense_path_tasks"] = int(st.trsm_dense_tasks)  # TSTRF + GESSM tasks solved on the matrix cores
    out["ssssm_dense_mfma"]["front_workgroups"] = int(st.ssssm_front_workgroups)      # dense-front kernel (pg_hip_front.h)
    out["ssssm_dense_mfma"]["general_workgroups"] = int(st.ssssm_general_workgroups)  # general MFMA kernel (pg_hip_dense.h)
    # class 5 by kernel: time of each kernel's launches (PROFILE on) and the flops its 16 x 16 x 16 products executed (COUNT_FLOPS on)
    out["ssssm_dense_mfma"]["front_kernel_ms"] = float(st.ssssm_kernel_ms[0])
    out["ssssm_dense_mfma"]["general_kernel_ms"] = float(st.ssssm_kernel_ms[1])
    out["ssssm_dense_mfma"]["front_flops_executed"] = float(st.ssssm_front_flops_executed)
    out["getrf"]["chase_launches"] = int(st.chase_launches)  # launches that carried a level's factorisations and its dense solves
    out["tstrf"]["chase_solves"] = int(st.chase_solves)
    return out
