"""Effect of the ordering inside separators on the 16 x 16 pieces of the factor blocks, from the symbolic pattern alone (CPU):
    PANGULU_AMD_SEPARATOR_ORDER=natural python tools/separator_order_eval.py 56 ; python tools/separator_order_eval.py 56 [fem27|shell|poisson3d] [nocoords]"""
import sys,os,time; sys.path.insert(0, '.')
import numpy as np, pangulu_amd as pa
from pangulu_amd import matrices as M
from tests.helpers import library_for, oracle_library
lib = library_for(oracle_library("r64"))
N=int(sys.argv[1]); nb=256
which=sys.argv[2] if len(sys.argv)>2 else "fem27"
mat={"fem27":lambda:M.fem27(N),"shell":lambda:M.shell(N,N),"poisson3d":lambda:M.poisson3d(N)}[which](); n,cp,ri,va,co=mat
if len(sys.argv)>3 and sys.argv[3]=="nocoords": co=None  # the graph-only ordering (separators ordered by pseudo-coordinates)
h = pa.pangulu_init(n,len(va),cp,ri,va,nb=nb,ordering="nd",coords=co,lib=lib,nthread=8)
info=h.info()
blocks={}
for brow,bcol,up,bcp,bri,bva in pa.owned_blocks(h):
    if brow==bcol: continue
    cols=np.repeat(np.arange(nb),np.diff(bcp.astype(np.int64)))
    m=np.zeros((16,16),bool)
    m[bri.astype(np.int64)>>4, cols>>4]=True
    blocks[(brow,bcol)]=(m,len(bri))
pa.pangulu_finalize(h)
Lcol={}; Urow={}
for (i,j) in blocks:
    if i>j: Lcol.setdefault(j,[]).append(i)
    else: Urow.setdefault(i,[]).append(j)
thr=(0.002*nb*nb)**2
nsteps=0; prods=0; tasks=0; struct=0.0
for k in Lcol:
    if k not in Urow: continue
    for i in Lcol[k]:
        A,na=blocks[(i,k)]
        for j in Urow[k]:
            if i!=j and (i,j) not in blocks: continue
            B,nbz=blocks[(k,j)]
            if na*nbz < thr: continue
            tasks+=1
            for tm in range(2):
                for tn in range(2):
                    a=A[8*tm:8*tm+8,:].astype(np.int64); b=B[:,8*tn:8*tn+8].astype(np.int64)
                    ar=a.sum(0); bc=b.sum(1)
                    live=(ar>0)&(bc>0)
                    nsteps+=int(live.sum()); prods+=int((ar*bc).sum())
pieces=sum(int(m.sum()) for m,_ in blocks.values()); nnzb=sum(c for _,c in blocks.values())
print("%s(%d) order=%s: symbolic nnz %d, flop %.4e, blocks %d, live 16x16 pieces %d (fill of a live piece %.1f%%), dense-mode tasks %d, live slab steps %d, live products %d (%.1f%% of the products of live steps), executed MFMA flops %.3e"%(
  which,N,os.environ.get("PANGULU_AMD_SEPARATOR_ORDER","kd"),info["symbolic_nnz"],info["flop"],len(blocks),pieces,100.0*nnzb/(pieces*256.0),tasks,nsteps,prods,100.0*prods/(64.0*nsteps),prods*8192.0))
