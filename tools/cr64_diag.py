"""poisson3d(N) CR64: residual of pangulu_gstrs and the device-side factor check side by side (which of the two phases is off?)
    python tools/cr64_diag.py N nb"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pangulu_amd as pa
from pangulu_amd import _lib, matrices as M
N = int(sys.argv[1]); nb = int(sys.argv[2])
import os
vt = os.environ.get("DIAG_VTYPE", "cr64")
lib = _lib.load(vt)
n, cp, ri, va, co = M.poisson3d(N, dtype=np.complex128, shift=0.5j) if vt == "cr64" else M.poisson3d(N)
b = M.rhs_of_ones(n, cp, ri, va)
import os
lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_COUNT_FLOPS, 0)
if os.environ.get("DIAG_PERMILLE"):
    lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_DENSE_THRESHOLD_PERMILLE, int(os.environ["DIAG_PERMILLE"]))
for kv in filter(None, os.environ.get("DIAG_OPT", "").split(",")):
    lib.pangulu_platform_0201001_set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
resets = int(os.environ.get("DIAG_RESETS", "0"))
h = pa.pangulu_init(n, len(va), cp, ri, va, nb=nb, vtype=vt, coords=co, nthread=32)
if resets:
    lib.pangulu_amd_snapshot(h.ref)
for i in range(resets + 1):
    t0 = time.time(); pa.pangulu_gstrf(h); t = time.time() - t0
    if i < resets:
        print("  pass %d factor check %.2e" % (i, pa.factor_check(h)), flush=True)
        lib.pangulu_amd_reset_numeric(h.ref)
fc = pa.factor_check(h)
x = pa.pangulu_gstrs(h, b)
print(vt, "poisson3d(%d) nb=%d n=%d: %.1f ms  factor check %.2e  residual %.2e  max|x-1| %.2e" % (N, nb, n, 1e3 * t, fc, M.relative_residual(n, cp, ri, va, x, b), abs(x - 1).max()), flush=True)
pa.pangulu_finalize(h)
