// pg_hip_block_solve.h -- device kernels of the level-scheduled block triangular solve (pangulu_gstrs on one rank) and of the
// device-side factor check.  Included inside the anonymous namespace of pg_hip_platform.hip.
#pragma once

// -----------------------------------------------------------------------------------------------------------------
// Level-scheduled block triangular solve for pangulu_gstrs on a single rank (pangulu_platform_0201001_block_trsv).
// The reference sweeps block row by block row with one spmv / sptrsv platform call per block on the CPU platform
// (src/pangulu_sptrsv.c:24-191); here block rows whose inputs are final form a LEVEL of the block dependency graph and one
// level is two launches: one workgroup per off-diagonal block subtracts  A(row, j) x_j  from the row's segment (floating-
// point atomics), then one wavefront per block row solves with the row's diagonal half in LDS and writes the finished
// segment.  Same per-block arithmetic as ...0100000.c:435-506 (spmv, unit-lower column sweep, upper row sweep
// with the PANGULU_SPTRSV_TOL clamp); sums across blocks arrive in a different order.
// -----------------------------------------------------------------------------------------------------------------
struct SolveBlkD
{
    const u32 *cp; // CSC
    const u16 *ri;
    const val_t *val;
    u32 bcol;
    u32 brow; // destination segment
};
struct SolveRowD
{
    u32 brow, nblk;
    unsigned long long first; // into the SolveBlkD array
    const u32 *dptr;          // diagonal half: lower = strictly-lower CSC column pointer, upper = CSR row pointer (diagonal first)
    const u16 *didx;
    const val_t *dval;
};

// x_row -= A(row, j) x_j for every off-diagonal block of the level: one workgroup per block (rows near the root of the
// tree have hundreds of blocks: a workgroup per row would walk them one after the other), floating-point atomics on
// the destination segment
__global__ __launch_bounds__(256) void block_trsv_gather_kernel(const SolveBlkD *__restrict__ blks, int nb, val_t *__restrict__ x)
{
    const SolveBlkD B = blks[blockIdx.x];
    const val_t *xj = x + (size_t)B.bcol * nb;
    val_t *xr = x + (size_t)B.brow * nb;
    const int sub = threadIdx.x >> 4, l16 = threadIdx.x & 15, nsub = blockDim.x >> 4;
    for (int c = sub; c < nb; c += nsub)
    {
        const u32 p0 = ptr0(B.cp, c), p1 = B.cp[c + 1];
        if (p0 == p1)
            continue;
        const val_t xc = xj[c];
        for (u32 p = p0 + l16; p < p1; p += 16)
        {
            const val_t m = v_mul(B.val[p], xc);
#ifdef PANGULU_COMPLEX
            v_atomic_add(&xr[B.ri[p]], val_t{-m.re, -m.im});
#else
            v_atomic_add(&xr[B.ri[p]], -m);
#endif
        }
    }
}

// y_dst += A x_src for a list of blocks (factor check: t = U 1, then y = L t): one workgroup per block, 16 lanes per
// column (CSC record) or row (CSR record: upper diagonal half), floating-point atomics on y
struct SpmvBlkD
{
    const u32 *ptr;
    const u16 *idx;
    const val_t *val;
    u32 src, dst;
    u32 csr, pad_;
};
__global__ __launch_bounds__(256) void block_spmv_add_kernel(const SpmvBlkD *__restrict__ blks, int nb, const val_t *__restrict__ x, val_t *__restrict__ y)
{
    const SpmvBlkD B = blks[blockIdx.x];
    const val_t *xs = x + (size_t)B.src * nb;
    val_t *yd = y + (size_t)B.dst * nb;
    const int sub = threadIdx.x >> 4, l16 = threadIdx.x & 15, nsub = blockDim.x >> 4;
    for (int c = sub; c < nb; c += nsub)
    {
        const u32 p0 = ptr0(B.ptr, c), p1 = B.ptr[c + 1];
        if (p0 == p1)
            continue;
        if (!B.csr)
        {
            const val_t xc = xs[c];
            for (u32 p = p0 + l16; p < p1; p += 16)
                v_atomic_add(&yd[B.idx[p]], v_mul(B.val[p], xc));
        }
        else
        {
            val_t part = v_make(0);
            for (u32 p = p0 + l16; p < p1; p += 16)
            {
                const val_t m = v_mul(B.val[p], xs[B.idx[p]]);
#ifdef PANGULU_COMPLEX
                part.re += m.re;
                part.im += m.im;
#else
                part += m;
#endif
            }
            v_atomic_add(&yd[c], part); // (16 partial sums per row)
        }
    }
}

// the diagonal halves of the level's block rows: one wavefront per row, the segment in LDS
template <bool UPPER>
__global__ __launch_bounds__(64) void block_trsv_level_kernel(const SolveRowD *__restrict__ rows, int nb, val_t *__restrict__ x)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    val_t *seg = reinterpret_cast<val_t *>(smem_raw);
    const SolveRowD R = rows[blockIdx.x];
    val_t *xr = x + (size_t)R.brow * nb;
    for (int i = threadIdx.x; i < nb; i += blockDim.x)
        seg[i] = xr[i];
    __syncthreads();
    // the diagonal half, by one wavefront (LDS operations of a wavefront complete in order: no barriers in the sweep)
    if (threadIdx.x < 64)
    {
        const int lane = threadIdx.x;
        if (!UPPER)
        {
            for (int c = 0; c < nb; c++)
            {
                const u32 p0 = ptr0(R.dptr, c), p1 = R.dptr[c + 1];
                if (p0 == p1)
                    continue;
                const val_t xc = seg[c];
                for (u32 p = p0 + lane; p < p1; p += 64)
                    seg[R.didx[p]] = v_submul(seg[R.didx[p]], R.dval[p], xc);
                wave_lds_fence();
            }
        }
        else
        {
            for (int r = nb - 1; r >= 0; r--)
            {
                const u32 b = R.dptr[r], e = R.dptr[r + 1];
                if (b == e)
                    continue;
#ifdef PANGULU_COMPLEX
                val_t part = v_make(0);
                for (u32 p = b + 1 + lane; p < e; p += 64)
                {
                    const val_t m = v_mul(R.dval[p], seg[R.didx[p]]);
                    part.re += m.re;
                    part.im += m.im;
                }
                for (int off = 32; off > 0; off >>= 1)
                {
                    part.re += __shfl_down(part.re, off, 64);
                    part.im += __shfl_down(part.im, off, 64);
                }
#else
                val_t part = 0;
                for (u32 p = b + 1 + lane; p < e; p += 64)
                    part += R.dval[p] * seg[R.didx[p]];
                for (int off = 32; off > 0; off >>= 1)
                    part += __shfl_down(part, off, 64);
#endif
                if (lane == 0)
                {
                    val_t d = R.dval[b];
                    const real_t dr = v_realpart(d);
                    if (!((dr < 0 ? -dr : dr) > (real_t)PANGULU_SPTRSV_TOL))
                        d = v_make((real_t)PANGULU_SPTRSV_TOL);
                    seg[r] = v_div(v_sub(seg[r], part), d);
                }
                wave_lds_fence();
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += blockDim.x)
        xr[i] = seg[i];
}

// ---- round 4: the same two launches per level, rebuilt around where their time went (fem27(112): 540 launches, 394 ms) --------
// The gather kernel walked a block column by column, sixteen lanes a column, every entry a floating-point atomic on the row's
// 256 words in HBM (hundreds of blocks of a row near the root contend for them): 98 % of its wave cycles waiting.  The level kernel
// swept a diagonal half column by column straight from HBM: nb dependent round trips.
//  * gather: the block's entries flat over the workgroup (coalesced loads, the column of an entry by bisection in an LDS copy of the
//    column pointers), products accumulated in LDS (ds_add_f64), ONE global atomic per touched row of the segment at the end;
//  * level: the diagonal half streams through LDS in chunks of `ch` columns (rows for the upper sweep), double-buffered: three
//    wavefronts fetch chunk k + 1 while the first one sweeps chunk k out of LDS -- a dependent step costs LDS round trips, not HBM ones.
__global__ __launch_bounds__(256) void block_trsv_gather_flat_kernel(const SolveBlkD *__restrict__ blks, int nb, val_t *__restrict__ x)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    val_t *acc = reinterpret_cast<val_t *>(smem_raw);
    val_t *xs = acc + nb;
    u32 *cp = reinterpret_cast<u32 *>(xs + nb);
    const SolveBlkD B = blks[blockIdx.x];
    const val_t *xj = x + (size_t)B.bcol * nb;
    val_t *xr = x + (size_t)B.brow * nb;
    const int tid = threadIdx.x;
    for (int i = tid; i < nb; i += 256)
    {
        acc[i] = v_make(0);
        xs[i] = xj[i];
    }
    for (int i = tid; i <= nb; i += 256)
        cp[i] = i == 0 ? 0u : B.cp[i];
    __syncthreads();
    const u32 nnz = cp[nb];
    for (u32 p = (u32)tid; p < nnz; p += 256)
    {
        // column of entry p: the last c with cp[c] <= p
        int lo = 0, hi = nb;
        while (hi - lo > 1)
        {
            const int mid = (lo + hi) >> 1;
            if (cp[mid] <= p)
                lo = mid;
            else
                hi = mid;
        }
        lds_atomic_sub(&acc[B.ri[p]], v_mul(B.val[p], xs[lo]));
    }
    __syncthreads();
    for (int i = tid; i < nb; i += 256)
        v_atomic_add(&xr[i], acc[i]);
}

template <bool UPPER>
__global__ __launch_bounds__(256) void block_trsv_level_chunked_kernel(const SolveRowD *__restrict__ rows, int nb, val_t *__restrict__ x, int ch)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const size_t cap = (size_t)ch * (size_t)nb; // entries a chunk can hold
    val_t *seg = reinterpret_cast<val_t *>(smem_raw);
    val_t *bv0 = seg + nb, *bv1 = bv0 + cap;
    u32 *ptr = reinterpret_cast<u32 *>(bv1 + cap);
    u16 *bi0 = reinterpret_cast<u16 *>(ptr + nb + 2), *bi1 = bi0 + cap;
    const SolveRowD R = rows[blockIdx.x];
    val_t *xr = x + (size_t)R.brow * nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < nb; i += 256)
        seg[i] = xr[i];
    for (int i = tid; i <= nb; i += 256)
        ptr[i] = (!UPPER && i == 0) ? 0u : R.dptr[i];
    __syncthreads();
    const int nchunk = (nb + ch - 1) / ch;
    // chunk k: columns [k ch, (k + 1) ch) of the lower half in ascending order; rows [nb - (k + 1) ch, nb - k ch) of the upper half, descending
    auto lo_of = [&](int k) -> int { return UPPER ? max(0, nb - (k + 1) * ch) : k * ch; };
    auto hi_of = [&](int k) -> int { return UPPER ? nb - k * ch : min(nb, (k + 1) * ch); };
    auto fetch = [&](int k, int first, int nthr)
    {
        val_t *bv = (k & 1) ? bv1 : bv0;
        u16 *bi = (k & 1) ? bi1 : bi0;
        const u32 p0 = ptr[lo_of(k)], p1 = ptr[hi_of(k)];
        for (u32 p = p0 + (u32)first; p < p1; p += (u32)nthr)
        {
            bv[p - p0] = R.dval[p];
            bi[p - p0] = R.didx[p];
        }
    };
    fetch(0, tid, 256);
    __syncthreads();
    for (int k = 0; k < nchunk; k++)
    {
        if (wave != 0)
        {
            if (k + 1 < nchunk)
                fetch(k + 1, tid - 64, 192);
        }
        else
        {
            const val_t *bv = (k & 1) ? bv1 : bv0;
            const u16 *bi = (k & 1) ? bi1 : bi0;
            const int c0 = lo_of(k), c1 = hi_of(k);
            const u32 base = ptr[c0];
            if (!UPPER)
            {
                for (int c = c0; c < c1; c++)
                {
                    const u32 p0 = ptr[c] - base, p1 = ptr[c + 1] - base;
                    if (p0 == p1)
                        continue;
                    const val_t xc = seg[c];
                    for (u32 p = p0 + lane; p < p1; p += 64)
                        seg[bi[p]] = v_submul(seg[bi[p]], bv[p], xc);
                    wave_lds_fence();
                }
            }
            else
            {
                for (int r = c1 - 1; r >= c0; r--)
                {
                    const u32 b = ptr[r] - base, e = ptr[r + 1] - base;
                    if (b == e)
                        continue;
#ifdef PANGULU_COMPLEX
                    val_t part = v_make(0);
                    for (u32 p = b + 1 + lane; p < e; p += 64)
                    {
                        const val_t m = v_mul(bv[p], seg[bi[p]]);
                        part.re += m.re;
                        part.im += m.im;
                    }
                    for (int off = 32; off > 0; off >>= 1)
                    {
                        part.re += __shfl_down(part.re, off, 64);
                        part.im += __shfl_down(part.im, off, 64);
                    }
#else
                    val_t part = 0;
                    for (u32 p = b + 1 + lane; p < e; p += 64)
                        part += bv[p] * seg[bi[p]];
                    for (int off = 32; off > 0; off >>= 1)
                        part += __shfl_down(part, off, 64);
#endif
                    if (lane == 0)
                    {
                        val_t d = bv[b];
                        const real_t dr = v_realpart(d);
                        if (!((dr < 0 ? -dr : dr) > (real_t)PANGULU_SPTRSV_TOL))
                            d = v_make((real_t)PANGULU_SPTRSV_TOL);
                        seg[r] = v_div(v_sub(seg[r], part), d);
                    }
                    wave_lds_fence();
                }
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < nb; i += 256)
        xr[i] = seg[i];
}
