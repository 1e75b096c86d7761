import sys,os,time; sys.path.insert(0,'/root/repo')
os.environ["PANGULU_AMD_ANALYSIS_ONLY"]="1"
import numpy as np, pangulu_amd as pa, ctypes
from pangulu_amd import matrices as M, _lib
from tests.helpers import library_for, oracle_library
lib = library_for(oracle_library("r64"))
cases = eval(sys.argv[1])
for name,mat,nb in cases:
    n,cp,ri,va,co = mat
    for N in (1,2,4,8):
        lib.pangulu_amd_test_set_analysis_ranks(N)
        t0=time.time()
        h = pa.pangulu_init(n,len(va),cp,ri,va,nb=nb,ordering="nd",coords=co,lib=lib,nthread=8)
        i=h.info()
        f=(ctypes.c_double*N)(); t=(ctypes.c_double*N)(); c=(ctypes.c_double*N)()
        lib.pangulu_amd_rank_model(h.ref,t,f,c)
        print("%s nb=%d N=%d: flop share %.3f time share %.3f T*sum %.3f ms T*(N) %.3f ms comm max %.3f ms sent %.2f GB cp %.3f ms (%d tasks) init %.1fs" % (
            name,nb,N,i["model_rank_flop_share"],i["model_rank_time_share"],1e3*i["model_ranks_tstar_sum"],1e3*i["model_ranks_tstar_max"],
            1e3*i["model_comm_seconds_max"],i["model_sent_bytes_total"]/1e9,1e3*i["model_critical_path"],i["model_critical_path_tasks"],time.time()-t0), flush=True)
        pa.pangulu_finalize(h)
    lib.pangulu_amd_test_set_analysis_ranks(1)
