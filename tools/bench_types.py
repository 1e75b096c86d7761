"""One GPU, every value type: gstrf time of a 3D problem at nb = 256 with the dense paths on (default) and off (dense thresholds
1001: sparse kernels only, what R32 / CR32 ran on before round 3).   python tools/bench_types.py [N] [types...]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import pangulu_amd as pa
from pangulu_amd import _lib, matrices as M
N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
types = sys.argv[2:] or ["r64", "r32", "cr64", "cr32"]
nb = 256
DT = {"r64": np.float64, "r32": np.float32, "cr64": np.complex128, "cr32": np.complex64}
for vtype in types:
    dt = DT[vtype]
    cplx = np.issubdtype(dt, np.complexfloating)
    lib = _lib.load(vtype)
    mat = M.poisson3d(N, dtype=dt, shift=0.5j if cplx else 0.0)
    n, cp, ri, va, co = mat
    b = M.rhs_of_ones(n, cp, ri, va)
    for dense in (True, False):
        lib.pangulu_amd_reset_options()
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_COUNT_FLOPS, 0)
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_DENSE_THRESHOLD_PERMILLE, 2 if dense else 1001)
        lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_TRSM_DENSE_PERMILLE, 5 if dense else 1001)
        pa.hip_stats(lib, reset=True)
        h = pa.pangulu_init(n, len(va), cp, ri, va, nb=nb, vtype=vtype, coords=co, nthread=32)
        lib.pangulu_amd_snapshot(h.ref)
        ts = []
        for i in range(3):
            t0 = time.time(); pa.pangulu_gstrf(h); ts.append(time.time() - t0)
            if i < 2:
                lib.pangulu_amd_reset_numeric(h.ref)
        x = pa.pangulu_gstrs(h, b)
        F = h.info()["flop"]
        real = (4 if cplx else 1) * F
        print("poisson3d(%d) %s nb=%d n=%d F=%.3e %s: %.1f ms -> %.0f GFLOP/s in the type's arithmetic (%.0f real GFLOP/s)  residual %.1e" % (
            N, vtype.upper(), nb, n, F, "dense paths" if dense else "sparse kernels only", min(ts) * 1e3, F / min(ts) / 1e9, real / min(ts) / 1e9,
            M.relative_residual(n, cp, ri, va, x, b)), flush=True)
        st = pa.hip_stats(lib)
        print("    launches/tasks per class (recorded at init): " + ", ".join("%s %d/%d" % (k, v["launches"], v["tasks"]) for k, v in st.items()) +
              "; dense solves %d; replayed %s" % (st["tstrf"]["dense_path_tasks"], h.info().get("replayed")), flush=True)
        pa.pangulu_finalize(h)
