"""Pattern statistics of the dense solves (CPU, no numeric phase): 16x16x16 tile products the dense TSTRF/GESSM issue
with strip-tile skipping only (what trsm_dense_f64_kernel does) and with the factor's tiles skipped as well.
python tools/trsm_stats.py nx ny"""
import sys, numpy as np
sys.path.insert(0, ".")
import pangulu_amd as pa
from pangulu_amd import _lib, matrices as M
from tests.helpers import oracle_library, select_platform
from pangulu_amd.solver import owned_blocks
nx, ny = int(sys.argv[1]), int(sys.argv[2])
nb = 256
lib = _lib.load("r64")
select_platform(lib, oracle_library("r64"))
n, cp, ri, va, co = M.shell(nx, ny)
h = pa.pangulu_init(n, len(va), cp, ri, va, nb=nb, coords=co, nthread=8)
info = h.info()
N = info["n_padded"]; NB = (N + nb - 1) // nb
dU = {}; dL = {}; Lb = []; Ub = []
for brow, bcol, up, cpb, rib, vab in owned_blocks(h):
    major = np.repeat(np.arange(nb, dtype=np.int64), np.diff(cpb.astype(np.int64)))
    minor = rib.astype(np.int64)
    T = np.zeros((16, 16), bool)
    if brow == bcol and up:
        T[major // 16, minor // 16] = True      # CSR: major = row, minor = col -> T[rowtile, coltile]
        dU[brow] = T
    else:
        T[minor // 16, major // 16] = True      # CSC: T[rowtile, coltile]
        if brow == bcol: dL[brow] = T
        elif brow > bcol: Lb.append((brow, bcol, len(minor), T))
        else: Ub.append((brow, bcol, len(minor), T))
dense = cur = both = 0; ntask = 0
thr = 0.01
for (i, k, nnz, X) in Lb:        # TSTRF: X U = B, strips = row tiles r, panels p = column tiles; needs U(q,p), q<p
    if nnz < thr * nb * nb: continue
    ntask += 1
    U = dU[k]
    for r in range(16):
        lv = X[r, :]
        if not lv.any(): continue
        for p in range(16):
            dense += p + 1
            if lv[p]:
                cur += int(lv[:p].sum()) + 1
                both += int((lv[:p] & U[:p, p]).sum()) + 1
    dense += 0
for (k, j, nnz, X) in Ub:        # GESSM: L X = B, strips = column tiles c, panels p = row tiles; needs L(p,q), q<p
    if nnz < thr * nb * nb: continue
    ntask += 1
    L = dL[k]
    for c in range(16):
        lv = X[:, c]
        if not lv.any(): continue
        for p in range(16):
            dense += p + 1
            if lv[p]:
                cur += int(lv[:p].sum()) + 1
                both += int((lv[:p] & L[p, :p]).sum()) + 1
print("dense-path solves", ntask, "tile products: all tiles of live strips", dense, "strip-tile skipping (now)", cur, "with factor tiles skipped too", both, "ratio", both / max(1, cur))
fill = [T.sum() / 136 for T in dU.values()]
print("diagonal upper halves: mean tile fill of the triangle %.3f" % (np.mean([np.triu(T).sum() / 136 for T in dU.values()])))
