import sys,json,time
sys.path.insert(0,".")
import pangulu_amd as pa
from pangulu_amd import _lib, matrices as M
lib=_lib.load("r64")
mat=M.shell(398,398)
n,cp,ri,va,co=mat
h=pa.pangulu_init(n,len(va),cp,ri,va,nb=256,coords=co,nthread=32)
lib.pangulu_amd_snapshot(h.ref)
lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_COUNT_FLOPS,0)
for t in [int(x) for x in sys.argv[1:]]:
    lib.pangulu_platform_0201001_set_option(_lib.HIP_OPT_DENSE_THRESHOLD_PERMILLE,t)
    ts=[]
    for i in range(3):
        t0=time.time(); pa.pangulu_gstrf(h); ts.append(time.time()-t0); lib.pangulu_amd_reset_numeric(h.ref)
    st=pa.hip_stats(lib,reset=True)
    print("threshold",t,"ms",[round(x*1e3,1) for x in ts], "GF/s %.0f"%(h.info()["flop"]/min(ts)/1e9), "dense tasks",st["ssssm_dense_mfma"]["tasks"]//3,"sparse",st["ssssm_sparse"]["tasks"]//3, flush=True)
