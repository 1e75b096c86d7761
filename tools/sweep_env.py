"""Sweep one environment variable (read by the scheduler at every pangulu_gstrf) on the bench workload:
python tools/sweep_env.py VAR v1 v2 ... [--size nx ny]"""
import os, sys, time
sys.path.insert(0, ".")
import pangulu_amd as pa
from pangulu_amd import _lib, matrices as M
args = sys.argv[1:]
size = (398, 398)
if "--size" in args:
    k = args.index("--size"); size = (int(args[k + 1]), int(args[k + 2])); args = args[:k] + args[k + 3:]
var, vals = args[0], args[1:]
lib = _lib.load("r64")
n, cp, ri, va, co = M.shell(*size)
h = pa.pangulu_init(n, len(va), cp, ri, va, nb=256, coords=co, nthread=32)
lib.pangulu_amd_snapshot(h.ref)
lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_COUNT_FLOPS, 0)
for v in vals:
    os.environ[var] = v
    ts = []
    for i in range(int(os.environ.get('SWEEP_REPS', '4'))):
        t0 = time.time(); pa.pangulu_gstrf(h); ts.append(time.time() - t0); lib.pangulu_amd_reset_numeric(h.ref)
    print(var, "=", v, "ms", [round(x * 1e3, 1) for x in ts], "min %.1f median %.1f" % (min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3), "GF/s %.0f" % (h.info()["flop"] / min(ts) / 1e9), "batches", h.info()["batches"], flush=True)
