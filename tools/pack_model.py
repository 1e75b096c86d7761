"""What would packing buy the MFMA update?  From the symbolic pattern alone (CPU): every dense-mode update task (A = L(i,k), B = U(k,j)) of a
factorisation, its live 16 x 16 pieces, and the cost of its slab steps under a simple model -- a step of a 128 x 128 tile costs F + W * (products
of the busiest wavefront) / 8 (F ~ 0.5 W) -- for
  now   : fixed 128 x 128 tiles of the 256 x 256 block, strided piece ownership (the kernel of round 3)
  pack  : the destination block's live row / column pieces (union over everything that updates it) compacted first, tiles of 8 x 8 pieces over
          the compact grid, operands gathered to match
    python tools/pack_model.py 40 [fem27|shell|poisson3d]"""
import sys, os
sys.path.insert(0, '.')
import numpy as np, pangulu_amd as pa
from pangulu_amd import matrices as M
from tests.helpers import library_for, oracle_library
lib = library_for(oracle_library("r64"))
N = int(sys.argv[1]); nb = 256
which = sys.argv[2] if len(sys.argv) > 2 else "fem27"
mat = {"fem27": lambda: M.fem27(N), "shell": lambda: M.shell(N, N), "poisson3d": lambda: M.poisson3d(N), "elastic3d": lambda: M.elastic3d(N)}[which]()
n, cp, ri, va, co = mat
h = pa.pangulu_init(n, len(va), cp, ri, va, nb=nb, ordering="nd", coords=co, lib=lib, nthread=8)
blocks = {}
for brow, bcol, up, bcp, bri, bva in pa.owned_blocks(h):
    cols = np.repeat(np.arange(nb), np.diff(bcp.astype(np.int64)))
    m = np.zeros((16, 16), bool)
    m[bri.astype(np.int64) >> 4, cols >> 4] = True
    key = (brow, bcol)
    if key in blocks:
        blocks[key] = (blocks[key][0] | m, blocks[key][1] + len(bri))
    else:
        blocks[key] = (m, len(bri))
pa.pangulu_finalize(h)
Lcol = {}; Urow = {}
for (i, j) in blocks:
    if i > j: Lcol.setdefault(j, []).append(i)
    elif i < j: Urow.setdefault(i, []).append(j)
thr = (0.002 * nb * nb) ** 2
F = float(os.environ.get("F_OVER_W", "0.5"))

def wave_busiest(rows, cols):
    """rows, cols: boolean arrays of 8 (pieces of the tile); strided ownership: wave (wr, wc) owns rows 2mi+wr, cols wc+4ni"""
    r = [int(rows[0::2].sum()), int(rows[1::2].sum())]
    c = [int(cols[w] + cols[w + 4]) for w in range(4)]
    return max(r) * max(c)

tot = {"now": 0.0, "pack": 0.0, "merge": 0.0}; steps = {"now": 0, "pack": 0, "merge": 0}; prods = 0; full_now = 0; full_pack = 0
queues = {}  # (i, j, tm, tn) -> list of (rows8, cols8) live records in queue order; prods = 0; full_now = 0; full_pack = 0
hist = np.zeros(65, np.int64)
# union of live pieces per destination
tasks = []
for k in Lcol:
    if k not in Urow: continue
    for i in Lcol[k]:
        A, na = blocks[(i, k)]
        for j in Urow[k]:
            if (i, j) not in blocks: continue
            B, nbz = blocks[(k, j)]
            if na * nbz < thr: continue
            tasks.append((i, j, k))
UR = {}; UC = {}
for (i, j, k) in tasks:
    A = blocks[(i, k)][0]; B = blocks[(k, j)][0]
    kl = A.any(0) & B.any(1)          # live slabs
    ar = A[:, kl].any(1); bc = B[kl, :].any(0)
    UR[(i, j)] = UR.get((i, j), np.zeros(16, bool)) | ar
    UC[(i, j)] = UC.get((i, j), np.zeros(16, bool)) | bc
for (i, j, k) in tasks:
    A = blocks[(i, k)][0]; B = blocks[(k, j)][0]
    ur = np.flatnonzero(UR[(i, j)]); uc = np.flatnonzero(UC[(i, j)])
    for s in range(16):
        a = A[:, s]; b = B[s, :]
        if not a.any() or not b.any(): continue
        prods += int(a.sum()) * int(b.sum())
        # now: 2 x 2 geometric tiles
        for tm in range(2):
            for tn in range(2):
                ra = a[8 * tm:8 * tm + 8]; cb = b[8 * tn:8 * tn + 8]
                if ra.any() and cb.any():
                    steps["now"] += 1
                    p = int(ra.sum()) * int(cb.sum()); hist[p] += 1
                    tot["now"] += F + wave_busiest(ra, cb) / 8.0
                    full_now += p == 64
                    queues.setdefault((i, j, tm, tn), []).append((ra.copy(), cb.copy()))
        # pack: compact grid of the destination
        ap = a[ur]; bp = b[uc]
        for tm in range(0, len(ur), 8):
            for tn in range(0, len(uc), 8):
                ra = np.zeros(8, bool); cb = np.zeros(8, bool)
                x = ap[tm:tm + 8]; y = bp[tn:tn + 8]
                ra[:len(x)] = x; cb[:len(y)] = y
                if ra.any() and cb.any():
                    steps["pack"] += 1
                    tot["pack"] += F + wave_busiest(ra, cb) / 8.0
                    full_pack += int(ra.sum()) * int(cb.sum()) == 64
# merge: consecutive records of a tile's queue share one LDS stage (8 A pieces + 8 B pieces of room) and one barrier
def wave_load(ra, cb):
    r = np.array([ra[0::2].sum(), ra[1::2].sum()]); c = np.array([cb[w] + cb[w + 4] for w in range(4)])
    return np.outer(r, c)
CAP = int(os.environ.get("CAP", "8"))
for q in queues.values():
    na = nbp = 0; load = np.zeros((2, 4)); open_ = False
    for ra, cb in q:
        a_, b_ = int(ra.sum()), int(cb.sum())
        if open_ and (na + a_ > CAP or nbp + b_ > CAP):
            steps["merge"] += 1; tot["merge"] += F + load.max() / 8.0
            na = nbp = 0; load = np.zeros((2, 4)); open_ = False
        na += a_; nbp += b_; load += wave_load(ra, cb); open_ = True
    if open_:
        steps["merge"] += 1; tot["merge"] += F + load.max() / 8.0
ideal = prods / 64.0
print("%s(%d): %d dense tasks, live products %d (= %.0f full steps of pure matrix-core time)" % (which, N, len(tasks), prods, ideal))
for kname in ("now", "pack", "merge"):
    print("  %-5s: %8d live steps, model time %.0f step-units (F = %.2f W) -> matrix cores busy %.1f %%; completely live steps %d" % (
        kname, steps[kname], tot[kname], F, 100.0 * ideal / tot[kname], full_now if kname == "now" else full_pack))
c = np.cumsum(hist[::-1])[::-1]
print("  live products per live step (now): " + " ".join("%d:%d" % (p, hist[p]) for p in range(65) if hist[p]))
