"""Pattern statistics of the factor blocks (CPU, no numeric phase): how much of the dense-mode work is structurally zero
at the granularity of 16-wide slabs?  python tools/slab_stats.py nx ny"""
import sys, numpy as np
sys.path.insert(0, ".")
import pangulu_amd as pa
from pangulu_amd import _lib, matrices as M
from tests.helpers import oracle_library, select_platform
nx, ny = int(sys.argv[1]), int(sys.argv[2])
nb = 256
lib = _lib.load("r64")
select_platform(lib, oracle_library("r64"))
n, cp, ri, va, co = M.shell(nx, ny)
h = pa.pangulu_init(n, len(va), cp, ri, va, nb=nb, coords=co, nthread=8)
info = h.info()
print("n", n, "padded", info["n_padded"], "symbolic nnz", info["symbolic_nnz"])
N = info["n_padded"]; NB = (N + nb - 1) // nb
bl = {}; bu = {}
from pangulu_amd.solver import owned_blocks
for brow, bcol, up, cpb, rib, vab in owned_blocks(h):
    major = np.repeat(np.arange(nb, dtype=np.int64), np.diff(cpb.astype(np.int64)))
    minor = rib.astype(np.int64)
    if brow == bcol:
        continue
    rows, cols = minor, major          # CSC
    rm = int(np.bitwise_or.reduce(1 << (rows // 16))) if len(rows) else 0
    cm = int(np.bitwise_or.reduce(1 << (cols // 16))) if len(cols) else 0
    # per 128-row half / 128-col half live masks as well
    mp = np.zeros(16, np.int64)
    np.bitwise_or.at(mp, cols // 16, 1 << (rows // 16))
    (bl if brow > bcol else bu)[brow * NB + bcol] = (len(rows), rm, cm, mp)
print("blocks L", len(bl), "U", len(bu))
pop = lambda x: bin(x).count("1")
# SSSSM tasks: for every k, L(i,k) i>k and U(k,j) j>k
from collections import defaultdict
Lcol = defaultdict(list); Urow = defaultdict(list)
for key in bl:
    i, k = divmod(key, NB)
    if i > k: Lcol[k].append(i)
for key in bu:
    k, j = divmod(key, NB)
    if j > k: Urow[k].append(j)
tot = heavy = 0; slabs_full = slabs_live = 0; trsm_d = 0; pairs = 0; thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.15
for k in range(NB):
    for i in Lcol[k]:
        ca, ra_mask, ca_mask, mpa = bl[i * NB + k]
        for j in Urow[k]:
            cb, rb_mask, cb_mask, mpb = bu[k * NB + j]
            tot += 1
            if (ca / nb / nb) * (cb / nb / nb) >= thr ** 2:
                heavy += 1
                slabs_full += 16
                slabs_live += pop(ca_mask & rb_mask)
                # 16x16 output tiles actually touched per K-slab: rows of A slab kk x columns of B having row slab kk
                for kk in range(16):
                    ra = pop(int(mpa[kk]))
                    if ra:
                        pairs += ra * sum(1 for c in range(16) if (int(mpb[c]) >> kk) & 1)
print("ssssm tasks", tot, "heavy", heavy, "K-slabs live fraction", slabs_live / max(1, slabs_full), "16x16x16 tile products live fraction", pairs / max(1, slabs_full * 256))
# TRSM dense: fill >= 10%: fraction of 16-row strips (TSTRF: rows of L block) that are non-empty, and leading-zero panels
st_full = st_live = 0
for key, (c, rm, cm, _m) in bl.items():
    i, k = divmod(key, NB)
    if i > k and c >= 0.10 * nb * nb:
        st_full += 16; st_live += pop(rm)
for key, (c, rm, cm, _m) in bu.items():
    k, j = divmod(key, NB)
    if j > k and c >= 0.10 * nb * nb:
        st_full += 16; st_live += pop(cm)
print("dense-solve strips live fraction", st_live / max(1, st_full), "strips", st_full)
