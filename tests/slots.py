"""Hand-built block descriptors for per-operator tests: what a reference host hands to the platform operators.

A slot (`pangulu_storage_slot_t`, include/pangulu_platform.h; reference src/pangulu_common.h:207-231 with GPU_OPEN)
points at one contiguous block record `[32 B header][values][colptr u32 x (nb+1)][rowidx u16 x nnz][pad 8]`
(+ `[csr->csc u32 x nnz][rowptr u32 x (nb+1)][colidx u16 x nnz][pad 8]` for lower blocks), once in host memory
and once in device memory (src/pangulu_communication.c:1290-1393, src/pangulu_storage.c:295-357).
The same slots are handed to the oracle's 0100000 operators (host fields) and to the HIP 0201001 operators (d_* fields).
"""
import ctypes

import numpy as np

from pangulu_amd import _lib


class Slot(ctypes.Structure):
    _fields_ = [
        ("brow_pos", ctypes.c_uint32), ("bcol_pos", ctypes.c_uint32),
        ("columnpointer", ctypes.c_void_p), ("rowindex", ctypes.c_void_p), ("value", ctypes.c_void_p),
        ("rowpointer", ctypes.c_void_p), ("columnindex", ctypes.c_void_p), ("idx_of_csc_value_for_csr", ctypes.c_void_p),
        ("data_status", ctypes.c_char), ("related_block", ctypes.c_void_p),
        ("is_upper", ctypes.c_int32), ("bin_id", ctypes.c_int32), ("slot_idx", ctypes.c_int32),
        ("task_queue", ctypes.c_void_p),
        ("d_columnpointer", ctypes.c_void_p), ("d_rowindex", ctypes.c_void_p), ("d_value", ctypes.c_void_p),
        ("d_rowpointer", ctypes.c_void_p), ("d_columnindex", ctypes.c_void_p), ("d_idx_of_csc_value_for_csr", ctypes.c_void_p),
    ]


class Task(ctypes.Structure):
    _fields_ = [
        ("row", ctypes.c_uint32), ("col", ctypes.c_uint32), ("kernel_id", ctypes.c_int16), ("task_level", ctypes.c_uint32),
        ("compare_flag", ctypes.c_int64), ("opdst", ctypes.c_void_p), ("op1", ctypes.c_void_p), ("op2", ctypes.c_void_p),
    ]


assert ctypes.sizeof(Slot) == 144 and ctypes.sizeof(Task) == 48
GETRF, TSTRF, GESSM, SSSSM = 1, 2, 3, 4


def _pad8(x):
    return (x + 7) & ~7


class Block:
    """One block record on the host (numpy buffer) and, when `lib` is the HIP library, on the device."""

    def __init__(self, nb, brow, bcol, is_upper, colptr, rowidx, values, dtype, hip_lib=None):
        self.nb, self.brow, self.bcol, self.is_upper = nb, brow, bcol, int(is_upper)
        self.dtype = np.dtype(dtype)
        nnz = int(colptr[nb])
        self.nnz = nnz
        sv = self.dtype.itemsize
        lower = (not is_upper) and brow >= bcol  # CSR view for every lower block (diagonal lower halves included)
        self.lower = lower
        first = _pad8(32 + sv * nnz + 4 * (nb + 1) + 2 * nnz)
        total = first + (_pad8(4 * nnz + 4 * (nb + 1) + 2 * nnz) if lower else 0)
        self.buf = np.zeros(total + 64, dtype=np.uint8)  # (host record; slack keeps views of empty arrays in range)
        self.total = total
        hdr = self.buf[:32]
        hdr[:8].view(np.uint64)[0] = nnz
        hdr[8:20].view(np.uint32)[:] = (brow, bcol, self.is_upper)
        self.off_val = 32
        self.off_cp = 32 + sv * nnz
        self.off_ri = self.off_cp + 4 * (nb + 1)
        self.values = self.buf[self.off_val:self.off_val + sv * nnz].view(self.dtype)
        self.colptr = self.buf[self.off_cp:self.off_cp + 4 * (nb + 1)].view(np.uint32)
        self.rowidx = self.buf[self.off_ri:self.off_ri + 2 * nnz].view(np.uint16)
        self.values[:] = values
        self.colptr[:] = colptr
        self.rowidx[:] = rowidx
        if lower:
            self.off_c2r = first
            self.off_rp = first + 4 * nnz
            self.off_ci = self.off_rp + 4 * (nb + 1)
            c2r = self.buf[self.off_c2r:self.off_c2r + 4 * nnz].view(np.uint32)
            rp = self.buf[self.off_rp:self.off_rp + 4 * (nb + 1)].view(np.uint32)
            ci = self.buf[self.off_ci:self.off_ci + 2 * nnz].view(np.uint16)
            # CSR view: entries sorted by (row, column); idx_of_csc_value_for_csr maps CSR position -> CSC position
            cols = np.repeat(np.arange(nb, dtype=np.int64), np.diff(np.asarray(colptr, dtype=np.int64)))
            rows = np.asarray(rowidx, dtype=np.int64)
            order = np.lexsort((cols, rows))
            c2r[:] = order
            ci[:] = cols[order]
            rp[0] = 0
            rp[1:] = np.cumsum(np.bincount(rows, minlength=nb))
        self.slot = Slot()
        s = self.slot
        s.brow_pos, s.bcol_pos, s.is_upper = brow, bcol, self.is_upper
        s.data_status = bytes([2])  # PANGULU_DATA_READY
        base = self.buf.ctypes.data
        s.value = base + self.off_val
        s.columnpointer = base + self.off_cp
        s.rowindex = base + self.off_ri
        if lower:
            s.idx_of_csc_value_for_csr = base + self.off_c2r
            s.rowpointer = base + self.off_rp
            s.columnindex = base + self.off_ci
        self.hip = hip_lib
        self.drec = None
        if hip_lib is not None:
            p = ctypes.c_void_p()
            hip_lib.pangulu_platform_0201001_malloc(ctypes.byref(p), total + 64)
            self.drec = p.value
            d = self.drec
            s.d_value = d + self.off_val
            if self.is_upper and brow == bcol:
                s.d_rowpointer = d + self.off_cp
                s.d_columnindex = d + self.off_ri
            else:
                s.d_columnpointer = d + self.off_cp
                s.d_rowindex = d + self.off_ri
            if lower:
                s.d_idx_of_csc_value_for_csr = d + self.off_c2r
                s.d_rowpointer = d + self.off_rp
                s.d_columnindex = d + self.off_ci
            self.upload()

    def upload(self):
        self.hip.pangulu_platform_0201001_memcpy(ctypes.c_void_p(self.drec), ctypes.c_void_p(self.buf.ctypes.data), self.total, 0)

    def download_values(self):
        """Device values -> a fresh numpy array (the host record is left alone)."""
        out = np.zeros(max(self.nnz, 1), dtype=self.dtype)
        if self.nnz:
            self.hip.pangulu_platform_0201001_memcpy(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(self.drec + self.off_val),
                                                     self.nnz * self.dtype.itemsize, 1)
        return out[:self.nnz]

    def free(self):
        if self.drec is not None:
            self.hip.pangulu_platform_0201001_free(ctypes.c_void_p(self.drec))
            self.drec = None

    def ref(self):
        return ctypes.byref(self.slot)

    def addr(self):
        return ctypes.addressof(self.slot)


def declare_platform(lib, pid):
    """ctypes signatures of the numeric operators of platform `pid` ("0201001" HIP, "0100000" oracle)."""
    P = ctypes.POINTER(Slot)
    vp = ctypes.c_void_p
    f = lambda n: getattr(lib, "pangulu_platform_%s_%s" % (pid, n))  # noqa: E731
    f("getrf").argtypes = [ctypes.c_uint16, P, ctypes.c_int]
    f("tstrf").argtypes = [ctypes.c_uint16, P, P, ctypes.c_int]
    f("gessm").argtypes = [ctypes.c_uint16, P, P, ctypes.c_int]
    f("ssssm").argtypes = [ctypes.c_uint16, P, P, P, ctypes.c_int]
    f("ssssm_batched").argtypes = [ctypes.c_uint16, ctypes.c_uint64, ctypes.POINTER(Task)]
    f("hybrid_batched").argtypes = [ctypes.c_uint16, ctypes.c_uint64, ctypes.POINTER(Task)]
    f("spmv").argtypes = [ctypes.c_uint16, P, vp, vp]
    f("vecadd").argtypes = [ctypes.c_int64, vp, vp]
    f("sptrsv").argtypes = [ctypes.c_uint16, P, vp, ctypes.c_int64]
    f("malloc").argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]
    f("memcpy").argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_uint]
    f("free").argtypes = [vp]
    f("synchronize").argtypes = []
    for n in ("getrf", "tstrf", "gessm", "ssssm", "ssssm_batched", "hybrid_batched", "spmv", "vecadd", "sptrsv", "malloc", "memcpy",
              "free", "synchronize"):
        f(n).restype = None
    return f


class BlockMatrix:
    """All block records of a small matrix (exported from pangulu_init on the oracle platform, before gstrf), rebuilt as
    hand-made slots for one platform, plus the reference's right-looking task order."""

    def __init__(self, records, nb, dtype, hip_lib=None):
        self.nb = nb
        self.blocks = {}  # (brow, bcol, is_upper) -> Block
        for brow, bcol, up, cp, ri, va in records:
            self.blocks[(brow, bcol, int(up))] = Block(nb, brow, bcol, up, cp, ri, va, dtype, hip_lib)
        self.nblk = 1 + max(k[0] for k in self.blocks)
        for k in range(self.nblk):
            lo, up = self.blocks[(k, k, 0)], self.blocks[(k, k, 1)]
            lo.slot.related_block = up.addr()
            up.slot.related_block = lo.addr()

    def get(self, brow, bcol):
        if brow == bcol:
            return self.blocks[(brow, bcol, 0)]
        return self.blocks.get((brow, bcol, 0))

    def tasks(self):
        """(kernel, dst, op1, op2) in the serial right-looking order a one-rank reference run executes level by level."""
        out = []
        n = self.nblk
        for k in range(n):
            d = self.blocks[(k, k, 1)]  # the reference passes either half (…0100000.c:62-70); use the upper one
            out.append((GETRF, d, None, None))
            below = [self.get(i, k) for i in range(k + 1, n) if self.get(i, k) is not None]
            right = [self.get(k, j) for j in range(k + 1, n) if self.get(k, j) is not None]
            for b in below:
                out.append((TSTRF, b, d, None))
            for b in right:
                out.append((GESSM, b, self.blocks[(k, k, 0)], None))
            for a in below:
                for b in right:
                    c = self.get(a.brow, b.bcol)
                    if c is not None:
                        out.append((SSSSM, c, a, b))
        return out

    def task_array(self, tasks):
        arr = (Task * len(tasks))()
        for i, (kid, dst, a, b) in enumerate(tasks):
            arr[i].kernel_id = kid
            arr[i].row, arr[i].col = dst.brow, dst.bcol
            arr[i].task_level = min(dst.brow, dst.bcol) if kid != SSSSM else a.bcol
            arr[i].opdst = dst.addr()
            arr[i].op1 = a.addr() if a is not None else None
            arr[i].op2 = b.addr() if b is not None else None
        return arr

    def free(self):
        for b in self.blocks.values():
            b.free()


def exported_records(mat, nb, vtype="r64", ordering="nd", user_perm=None):
    """Block records (patterns closed under fill, values = A on its pattern, 0 on fill) as the host builds them.
    `user_perm` (perm[new] = old): the permutation to use instead of computing an ordering -- a fixture that carries its
    permutation pins the operators, not the ordering code."""
    import pangulu_amd as pa

    from .helpers import library_for, oracle_library

    n, cp, ri, va, coords = mat
    lib = library_for(oracle_library(vtype), vtype)
    if user_perm is not None:
        h = pa.pangulu_init(n, len(va), cp, ri, va, nb=nb, vtype=vtype, ordering="user", user_perm=user_perm, lib=lib)
    else:
        h = pa.pangulu_init(n, len(va), cp, ri, va, nb=nb, vtype=vtype, ordering=ordering, coords=coords if ordering == "nd" else None, lib=lib)
    recs = list(pa.owned_blocks(h))
    pa.pangulu_finalize(h)
    return recs
