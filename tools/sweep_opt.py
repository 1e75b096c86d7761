"""Sweep one back-end option on the bench workload: python tools/sweep_opt.py <option id> v1 v2 ... [--size nx ny]"""
import os, sys, time
sys.path.insert(0, ".")
import pangulu_amd as pa
from pangulu_amd import _lib, matrices as M
args = sys.argv[1:]
size = (398, 398)
if "--size" in args:
    k = args.index("--size"); size = (int(args[k + 1]), int(args[k + 2])); args = args[:k] + args[k + 3:]
fixed = []
while "--set" in args:  # --set id=value: other options held fixed during the sweep
    k = args.index("--set"); o, v = args[k + 1].split("="); fixed.append((int(o), int(v))); args = args[:k] + args[k + 2:]
opt = int(args[0]); vals = [int(v) for v in args[1:]]
lib = _lib.load("r64")
n, cp, ri, va, co = M.shell(*size)
h = pa.pangulu_init(n, len(va), cp, ri, va, nb=256, coords=co, nthread=32)
lib.pangulu_amd_snapshot(h.ref)
lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_COUNT_FLOPS, 0)
for o, v in fixed:
    lib.pangulu_platform_0201001_set_option(o, v)
for v in vals:
    lib.pangulu_platform_0201001_set_option(opt, v)
    ts = []
    for i in range(3):
        t0 = time.time(); pa.pangulu_gstrf(h); ts.append(time.time() - t0); lib.pangulu_amd_reset_numeric(h.ref)
        if os.environ.get('PANGULU_HIP_HOST_TIMING'):
            print('host sched s', h.info()['time_numeric_host_sched'], flush=True); pa.hip_stats(lib, reset=True)
    st = pa.hip_stats(lib, reset=True)
    print("option", opt, "=", v, "ms", [round(x * 1e3, 1) for x in ts], "GF/s %.0f" % (h.info()["flop"] / min(ts) / 1e9),
          "ssssm dense/sparse", st["ssssm_dense_mfma"]["tasks"] // 3, st["ssssm_sparse"]["tasks"] // 3, "trsm dense", st["tstrf"]["dense_path_tasks"] // 3, flush=True)
