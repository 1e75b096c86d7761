"""BASELINE config 5 class on one GPU: complex-shifted 3D Poisson (CR64), gstrf time with the updates on the matrix cores
(default) and on the sparse kernels only (dense threshold 1001).   python tools/bench_cr64.py [N] [nb]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import pangulu_amd as pa
from pangulu_amd import _lib, matrices as M
N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 256
lib = _lib.load("cr64")
mat = M.poisson3d(N, dtype=np.complex128, shift=0.5j)
n, cp, ri, va, co = mat
b = M.rhs_of_ones(n, cp, ri, va)
for permille in (2, 1001):  # (2 = the back-end's default)
    lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_DENSE_THRESHOLD_PERMILLE, permille)
    lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_COUNT_FLOPS, 0)
    h = pa.pangulu_init(n, len(va), cp, ri, va, nb=nb, vtype="cr64", coords=co, nthread=32)
    lib.pangulu_amd_snapshot(h.ref)
    ts = []
    for i in range(3):
        t0 = time.time(); pa.pangulu_gstrf(h); ts.append(time.time() - t0)
        if i < 2:
            lib.pangulu_amd_reset_numeric(h.ref)
    fc = pa.factor_check(h)
    x = pa.pangulu_gstrs(h, b)
    res = M.relative_residual(n, cp, ri, va, x, b)
    assert fc < 1e-10 and res < 1e-10, (fc, res)  # a timing of wrong factors is not a measurement
    F = h.info()["flop"]
    print("poisson3d(%d) CR64 nb=%d n=%d F=%.3e dense_permille=%d: %.1f ms  -> %.0f GFLOP/s (structural count; a complex multiply-add is 8 real flops: x4 = %.0f real GFLOP/s)  residual %.1e  factor check %.1e" % (
        N, nb, n, F, permille, min(ts) * 1e3, F / min(ts) / 1e9, 4 * F / min(ts) / 1e9, res, fc), flush=True)
    pa.pangulu_finalize(h)
