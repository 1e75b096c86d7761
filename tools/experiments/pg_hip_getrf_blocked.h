// pg_hip_getrf_blocked.h -- round 1's dense GETRF kernels (LDS-blocked, and with look-ahead inside the block); selected by
// PANGULU_HIP_GETRF_TILED=0.  Included by pg_hip_platform.hip after the pattern-driven kernels, before pg_hip_getrf_tiled.h.
#pragma once

// -----------------------------------------------------------------------------------------------------------------
// GETRF, blocked (R64, nb a multiple of 16 up to 256).  The kernel above pays two L2 round trips per pivot
// (~1.5 us x nb): it is latency-bound whatever the fill.  This one keeps the active 16-column panel and the matching
// 16-row strip of U in LDS, eliminates inside them (LDS latency only), and applies the panel's rank-16 update to the
// trailing block on the f64 matrix cores straight from those LDS images:
//   for each panel j0:  P = D[j0:, j0:j0+16] (LDS, column-major)   S = D[j0:j0+16, j0+16:] (LDS, row-major)
//       16 pivots: scale L(:,k); rank-1 update of the rest of P and of S          (wavefront per column / row)
//       write P and S back;  D[j0+16:, j0+16:] -= P_lower * S                      (v_mfma_f64_16x16x4_f64)
// The dense image is zero outside the pattern; structural zeros make exact no-ops, so the factors equal the sparse
// algorithm's on the pattern.  Every entry still receives its updates in ascending pivot order.
// The trailing product is formed transposed (A operand = -S^T, B operand = P^T) so that each accumulator register
// maps to 16 consecutive rows of one column of D: loads and stores of the trailing block are 128-byte segments.
// -----------------------------------------------------------------------------------------------------------------
#if defined(PG_DENSE_PANELS)
#define GETRF_PANEL 16
#define GETRF_BLOCKED_ROWS 256 // one row thread per row: nb <= 256

// (owner_of and GETRF_STAMP: pg_hip_getrf_tiled.h, which has to be included first)

// THREADS = 1024: sixteen wavefronts, the whole register file of the CU (fastest for a block on its own).
// THREADS = 512: eight wavefronts capped at 128 registers -- half of the CU stays free, so the update and densify
// workgroups of a look-ahead batch run on the same CUs beside a launch that has a diagonal block for every CU.
template <int GETRF_BLOCKED_THREADS>
__global__ __launch_bounds__(GETRF_BLOCKED_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void getrf_blocked_f64_kernel(const GetrfTaskD *__restrict__ tasks, int nb,
                                                                                  unsigned long long *flop_counter,
                                                                                  unsigned long long *dbg)
{
    unsigned long long stamp_ = dbg ? __builtin_amdgcn_s_memtime() : 0;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int ldp = nb + 2; // leading dimensions padded by one 16-byte slot: MFMA operand reads stay conflict-free
    double *P = reinterpret_cast<double *>(smem_raw); // P[c * ldp + r]: column c (0..15) of the panel, row r (absolute)
    double *S = P + GETRF_PANEL * ldp;                // S[k * ldp + c]: row k (0..15) of the strip, column c (absolute)
    double *Rb = S + GETRF_PANEL * ldp;               // Rb[kk * 16 + c]: pivot row kk of the panel, published per step
    u32 *sLcp = reinterpret_cast<u32 *>(Rb + GETRF_PANEL * GETRF_PANEL); // column pointer of the lower half (nb + 1 entries)
    u32 *sUrp = sLcp + nb + 1;                                           // row pointer of the upper half
    const GetrfTaskD T = tasks[blockIdx.x];
    double *__restrict__ D = reinterpret_cast<double *>(T.dense);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwaves = GETRF_BLOCKED_THREADS / 64;

    // dense image: zero, then scatter both halves (unless the caller hands over a current dense mirror)
    if (!T.preloaded)
    {
        for (int i = tid; i < nb * nb / 2; i += GETRF_BLOCKED_THREADS)
            reinterpret_cast<double2 *>(D)[i] = make_double2(0.0, 0.0);
    }
    for (int i = tid; i <= nb; i += GETRF_BLOCKED_THREADS)
    {
        sLcp[i] = T.lcp[i];
        sUrp[i] = T.urp[i];
    }
    __syncthreads();
    // scatter / gather run flat over the nonzeros (coalesced, many loads in flight); the owning column (row) of a
    // position comes from a binary search in the LDS copy of the pointer array
    const u32 nnzL = sLcp[nb], nnzU = sUrp[nb];
    unsigned long long ops = 0;
    if (!T.preloaded)
    {
        for (u32 p = tid; p < nnzL; p += GETRF_BLOCKED_THREADS)
            D[(size_t)owner_of(sLcp, nb, p) * nb + T.lri[p]] = T.lval[p];
        for (u32 p = tid; p < nnzU; p += GETRF_BLOCKED_THREADS)
            D[(size_t)T.uci[p] * nb + owner_of(sUrp, nb, p)] = T.uval[p];
    }
    for (int c = tid; c < nb; c += GETRF_BLOCKED_THREADS)
    {
        // structural flop count of the sparse algorithm (what the reference counts, src/pangulu_kernel_interface.c:4-82)
        const u32 nl = sLcp[c + 1] - sLcp[c], nu = sUrp[c + 1] - sUrp[c];
        if (nu > 0)
            ops += (unsigned long long)nl * (1ull + 2ull * (nu - 1));
    }
    __syncthreads();
    GETRF_STAMP(0)

    for (int j0 = 0; j0 < nb; j0 += GETRF_PANEL)
    {
        const int jt = j0 + GETRF_PANEL; // first trailing row/column
        // ---- panel: thread t < nb - j0 owns row j0 + t of the 16 panel columns in registers ---------------------
        // Per pivot the owner of the pivot row publishes it through LDS (one barrier), every row below scales its
        // own L entry and updates its own 15 registers: no LDS traffic besides the 16-value broadcast.
        const int myrow = j0 + tid;
        const bool row_thread = tid < GETRF_BLOCKED_ROWS && myrow < nb;
        double x[GETRF_PANEL];
        if (row_thread)
        {
#pragma unroll
            for (int c = 0; c < GETRF_PANEL; c++)
                x[c] = D[(size_t)(j0 + c) * nb + myrow];
        }
        GETRF_STAMP(1)
#pragma unroll
        for (int kk = 0; kk < GETRF_PANEL; kk++)
        {
            if (tid == kk)
            {
#pragma unroll
                for (int c = 0; c < GETRF_PANEL; c++)
                    Rb[kk * GETRF_PANEL + c] = x[c];
            }
            __syncthreads();
            const int k = j0 + kk;
            if (sLcp[k] != sLcp[k + 1] && row_thread && myrow > k && x[kk] != 0.0)
            {
                const double l = x[kk] / clamp_pivot(Rb[kk * GETRF_PANEL + kk]);
                x[kk] = l;
#pragma unroll
                for (int c = 0; c < GETRF_PANEL; c++)
                    if (c > kk)
                        x[c] = x[c] - l * Rb[kk * GETRF_PANEL + c];
            }
        }
        GETRF_STAMP(2)
        if (row_thread)
        {
#pragma unroll
            for (int c = 0; c < GETRF_PANEL; c++)
            {
                D[(size_t)(j0 + c) * nb + myrow] = x[c];
                P[c * ldp + myrow] = x[c];
            }
        }
        __syncthreads();
        GETRF_STAMP(3)
        // ---- strip: thread t < nb - jt owns column jt + t of the 16 strip rows; forward substitution with the unit
        // lower 16 x 16 tile L11 read (broadcast) from the panel image ------------------------------------------------
        if (tid < nb - jt)
        {
            const int c = jt + tid;
            double s[GETRF_PANEL];
            const double2 *src = reinterpret_cast<const double2 *>(D + (size_t)c * nb + j0);
#pragma unroll
            for (int q = 0; q < GETRF_PANEL / 2; q++)
            {
                const double2 v = src[q];
                s[2 * q] = v.x;
                s[2 * q + 1] = v.y;
            }
#pragma unroll
            for (int kk = 0; kk < GETRF_PANEL; kk++)
            {
                if (s[kk] != 0.0)
                {
#pragma unroll
                    for (int rr = 0; rr < GETRF_PANEL; rr++)
                        if (rr > kk)
                            s[rr] = s[rr] - P[kk * ldp + j0 + rr] * s[kk];
                }
            }
            double2 *dst = reinterpret_cast<double2 *>(D + (size_t)c * nb + j0);
#pragma unroll
            for (int q = 0; q < GETRF_PANEL / 2; q++)
                dst[q] = make_double2(s[2 * q], s[2 * q + 1]);
#pragma unroll
            for (int kk = 0; kk < GETRF_PANEL; kk++)
                S[kk * ldp + c] = s[kk];
        }
        __syncthreads();
        GETRF_STAMP(4)
        // ---- trailing update on the matrix cores ---------------------------------------------------------------
        const int mt = (nb - jt) / 16; // trailing tiles per dimension
        const int l15 = lane & 15, l4 = lane >> 4;
        // a wavefront takes 32 x 32 macro tiles (2 x 2 MFMA tiles sharing their operands): 16 accumulator loads in
        // flight per pass instead of 4, half the LDS operand reads per flop
        const int mm = (mt + 1) / 2;
        for (int mtile = wave; mtile < mm * mm; mtile += nwaves)
        {
            const int ri = (mtile % mm) * 2, ci = (mtile / mm) * 2;
            const int r0 = jt + ri * 16, c0 = jt + ci * 16;
            const bool hr = ri + 1 < mt, hc = ci + 1 < mt; // second row / column of tiles exists
            const int r1 = hr ? r0 + 16 : r0, c1 = hc ? c0 + 16 : c0;
            // operands: A[i = l15][k = l4] = -U(k, c+i);  B[k = l4][j = l15] = L(r+j, k)
            double a0[4], a1[4], b0[4], b1[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
            {
                a0[q] = -S[(q * 4 + l4) * ldp + c0 + l15];
                a1[q] = -S[(q * 4 + l4) * ldp + c1 + l15];
                b0[q] = P[(q * 4 + l4) * ldp + r0 + l15];
                b1[q] = P[(q * 4 + l4) * ldp + r1 + l15];
            }
            const bool za0 = !__any((a0[0] != 0.0) | (a0[1] != 0.0) | (a0[2] != 0.0) | (a0[3] != 0.0));
            const bool za1 = !hc || !__any((a1[0] != 0.0) | (a1[1] != 0.0) | (a1[2] != 0.0) | (a1[3] != 0.0));
            const bool zb0 = !__any((b0[0] != 0.0) | (b0[1] != 0.0) | (b0[2] != 0.0) | (b0[3] != 0.0));
            const bool zb1 = !hr || !__any((b1[0] != 0.0) | (b1[1] != 0.0) | (b1[2] != 0.0) | (b1[3] != 0.0));
            // tile (x, y) = rows r_x, columns c_y; skipped when its L rows or U columns are all zero (uniform)
            const bool d00 = !(zb0 || za0), d10 = !(zb1 || za0), d01 = !(zb0 || za1), d11 = !(zb1 || za1);
            v4f64 t00 = {0.0, 0.0, 0.0, 0.0}, t10 = t00, t01 = t00, t11 = t00;
            // accumulator register g of lane l is D(r + l15, c + l4 + 4g)
#pragma unroll
            for (int g = 0; g < 4; g++)
            {
                if (d00)
                    t00[g] = D[(size_t)(c0 + l4 + 4 * g) * nb + r0 + l15];
                if (d10)
                    t10[g] = D[(size_t)(c0 + l4 + 4 * g) * nb + r1 + l15];
                if (d01)
                    t01[g] = D[(size_t)(c1 + l4 + 4 * g) * nb + r0 + l15];
                if (d11)
                    t11[g] = D[(size_t)(c1 + l4 + 4 * g) * nb + r1 + l15];
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
            {
                if (d00)
                    t00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b0[q], t00, 0, 0, 0);
                if (d10)
                    t10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b1[q], t10, 0, 0, 0);
                if (d01)
                    t01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b0[q], t01, 0, 0, 0);
                if (d11)
                    t11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b1[q], t11, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; g++)
            {
                if (d00)
                    D[(size_t)(c0 + l4 + 4 * g) * nb + r0 + l15] = t00[g];
                if (d10)
                    D[(size_t)(c0 + l4 + 4 * g) * nb + r1 + l15] = t10[g];
                if (d01)
                    D[(size_t)(c1 + l4 + 4 * g) * nb + r0 + l15] = t01[g];
                if (d11)
                    D[(size_t)(c1 + l4 + 4 * g) * nb + r1 + l15] = t11[g];
            }
        }
        __syncthreads();
        GETRF_STAMP(5)
    }

    if (T.defer_gather)
    {
        // the factors stay in the dense image; the diagonal tiles are saved behind the mirror (values + occupancy map)
        // because diag_tile_inverse_kernel replaces them by their inverses before the sparsify job reads the image
        double *__restrict__ saved = D + (size_t)nb * nb + MIRROR_MAP_BYTES / sizeof(double);
        for (int i = tid; i < nb * 16; i += GETRF_BLOCKED_THREADS)
        {
            const int pt = i >> 8, cc = (i >> 4) & 15, rr = i & 15;
            saved[i] = D[(size_t)(16 * pt + cc) * nb + 16 * pt + rr];
        }
    }
    else
    {
        // gather the factors back into the sparse record: four entries per thread and pass, so that the index loads, the
        // searches and the reads of D of different entries overlap (one entry at a time is a chain of three dependent L2
        // round trips per entry: 56 of the kernel's 330 us)
        constexpr int GU = 4;
        for (u32 p0 = tid; p0 < nnzL; p0 += GU * GETRF_BLOCKED_THREADS)
        {
            u32 r[GU];
            double v[GU];
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * GETRF_BLOCKED_THREADS;
                r[u] = p < nnzL ? T.lri[p] : 0u;
            }
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * GETRF_BLOCKED_THREADS;
                v[u] = p < nnzL ? D[(size_t)owner_of(sLcp, nb, p) * nb + r[u]] : 0.0;
            }
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * GETRF_BLOCKED_THREADS;
                if (p < nnzL)
                    T.lval[p] = v[u];
            }
        }
        for (u32 p0 = tid; p0 < nnzU; p0 += GU * GETRF_BLOCKED_THREADS)
        {
            u32 c[GU];
            double v[GU];
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * GETRF_BLOCKED_THREADS;
                c[u] = p < nnzU ? T.uci[p] : 0u;
            }
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * GETRF_BLOCKED_THREADS;
                v[u] = p < nnzU ? D[(size_t)c[u] * nb + owner_of(sUrp, nb, p)] : 0.0;
            }
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * GETRF_BLOCKED_THREADS;
                if (p < nnzU)
                    T.uval[p] = v[u];
            }
        }
    }
    __syncthreads();
    GETRF_STAMP(6)
    if (T.invert_tiles)
    {
        // one wavefront per diagonal tile, 16 x 17 doubles of the (now free) panel image each
        double(*Tw)[17] = reinterpret_cast<double(*)[17]>(smem_raw) + wave * 16;
        for (int p0 = 0; p0 < nb / 16; p0 += nwaves)
            invert_diag_tile(D, nb, p0 + wave, Tw, lane, p0 + wave < nb / 16, []()
                             { __syncthreads(); });
    }
    ops = wave_sum(ops);
    if (lane == 0 && ops)
        atomicAdd(flop_counter, ops);
}
// -----------------------------------------------------------------------------------------------------------------
// Blocked GETRF with look-ahead inside the block (PANGULU_HIP_GETRF_LOOKAHEAD=1): while twelve wavefronts apply panel j
// to the trailing block, the other four first update the tiles panel j+1 and its strip consist of, then eliminate panel
// j+1 and solve its strip -- the two latency-bound phases of a panel step run beside the trailing update of the previous
// one.  The four synchronise among themselves through an LDS counter.
// -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void getrf_lookahead_f64_kernel(const GetrfTaskD *__restrict__ tasks, int nb,
                                                                                  unsigned long long *flop_counter,
                                                                                  unsigned long long *dbg)
{
    unsigned long long stamp_ = dbg ? __builtin_amdgcn_s_memtime() : 0;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int ldp = nb + 2; // leading dimensions padded by one 16-byte slot: MFMA operand reads stay conflict-free
    // two images of the panel and of the strip: while the trailing update reads the current ones, the look-ahead
    // wavefronts build the next ones
    double *Pb = reinterpret_cast<double *>(smem_raw);  // Pb[(buf * 16 + c) * ldp + r]: column c of the panel, row r (absolute)
    double *Sb = Pb + 2 * GETRF_PANEL * ldp;            // Sb[(buf * 16 + k) * ldp + c]: row k of the strip, column c (absolute)
    double *Rb = Sb + 2 * GETRF_PANEL * ldp;            // Rb[kk * 16 + c]: pivot row kk of the panel, published per step
    u32 *sLcp = reinterpret_cast<u32 *>(Rb + GETRF_PANEL * GETRF_PANEL); // column pointer of the lower half (nb + 1 entries)
    u32 *sUrp = sLcp + nb + 1;                                           // row pointer of the upper half
    unsigned *la_count = sUrp + nb + 1;                                  // arrivals at the look-ahead wavefronts' own barrier
    const GetrfTaskD T = tasks[blockIdx.x];
    double *__restrict__ D = reinterpret_cast<double *>(T.dense);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwaves = 1024 / 64;

    // dense image: zero, then scatter both halves (unless the caller hands over a current dense mirror)
    if (!T.preloaded)
    {
        for (int i = tid; i < nb * nb / 2; i += 1024)
            reinterpret_cast<double2 *>(D)[i] = make_double2(0.0, 0.0);
    }
    for (int i = tid; i <= nb; i += 1024)
    {
        sLcp[i] = T.lcp[i];
        sUrp[i] = T.urp[i];
    }
    __syncthreads();
    // scatter / gather run flat over the nonzeros (coalesced, many loads in flight); the owning column (row) of a
    // position comes from a binary search in the LDS copy of the pointer array
    const u32 nnzL = sLcp[nb], nnzU = sUrp[nb];
    unsigned long long ops = 0;
    if (!T.preloaded)
    {
        for (u32 p = tid; p < nnzL; p += 1024)
            D[(size_t)owner_of(sLcp, nb, p) * nb + T.lri[p]] = T.lval[p];
        for (u32 p = tid; p < nnzU; p += 1024)
            D[(size_t)T.uci[p] * nb + owner_of(sUrp, nb, p)] = T.uval[p];
    }
    for (int c = tid; c < nb; c += 1024)
    {
        // structural flop count of the sparse algorithm (what the reference counts, src/pangulu_kernel_interface.c:4-82)
        const u32 nl = sLcp[c + 1] - sLcp[c], nu = sUrp[c + 1] - sUrp[c];
        if (nu > 0)
            ops += (unsigned long long)nl * (1ull + 2ull * (nu - 1));
    }
    __syncthreads();
    GETRF_STAMP(0)


    constexpr int LA = 4; // look-ahead wavefronts: all row threads (nb <= 256) and all strip threads live in them
    if (tid == 0)
        *la_count = 0;
    unsigned la_target = 0;
    // barrier of the LA look-ahead wavefronts only (the others are busy with the trailing update and must not be held
    // up): arrivals are counted in LDS.  LDS operations of a wavefront execute in order, so what a wavefront wrote
    // before it arrived is visible to whoever sees its arrival; vmcnt(0) orders its global stores the same way.
    auto la_barrier = [&]()
    {
        la_target += LA;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0)
            atomicAdd(la_count, 1u);
        while (*(volatile unsigned *)la_count < la_target)
            __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    const int l15 = lane & 15, l4 = lane >> 4;

    // ---- panel j0 into image `buf`: thread t < nb - j0 owns row j0 + t of the 16 panel columns in registers.  The 16 pivot
    // rows are rows of wavefront 0: it eliminates its 64 rows on its own, pivot row by pivot row, broadcasting each from its
    // lane with v_readlane (no LDS, no barrier) and publishing it for the others; after ONE barrier (`sync`) the other row
    // wavefronts run the same 16 steps on their rows from the published rows.  Same operations in the same order per
    // row as a barrier per pivot, a sixteenth of the barriers -- which matters here, where a barrier is an LDS counter.
    auto panel = [&](int j0, int buf, auto sync)
    {
        double *P = Pb + (size_t)buf * GETRF_PANEL * ldp;
        const int myrow = j0 + tid;
        const bool row_thread = tid < GETRF_BLOCKED_ROWS && myrow < nb;
        double x[GETRF_PANEL];
#pragma unroll
        for (int c = 0; c < GETRF_PANEL; c++)
            x[c] = row_thread ? D[(size_t)(j0 + c) * nb + myrow] : 0.0;
        if (wave == 0)
        {
#pragma unroll
            for (int kk = 0; kk < GETRF_PANEL; kk++)
            {
                double u[GETRF_PANEL]; // pivot row kk (wavefront-uniform)
#pragma unroll
                for (int c = 0; c < GETRF_PANEL; c++)
                    if (c >= kk)
                    {
                        union
                        {
                            double d;
                            int w[2];
                        } v;
                        v.d = x[c];
                        v.w[0] = __builtin_amdgcn_readlane(v.w[0], kk);
                        v.w[1] = __builtin_amdgcn_readlane(v.w[1], kk);
                        u[c] = v.d;
                    }
                if (lane == kk)
                {
#pragma unroll
                    for (int c = 0; c < GETRF_PANEL; c++)
                        Rb[kk * GETRF_PANEL + c] = x[c];
                }
                const int k = j0 + kk;
                if (sLcp[k] != sLcp[k + 1] && row_thread && myrow > k && x[kk] != 0.0)
                {
                    const double l = x[kk] / clamp_pivot(u[kk]);
                    x[kk] = l;
#pragma unroll
                    for (int c = 0; c < GETRF_PANEL; c++)
                        if (c > kk)
                            x[c] = x[c] - l * u[c];
                }
            }
        }
        sync();
        if (wave != 0 && row_thread)
        {
#pragma unroll
            for (int kk = 0; kk < GETRF_PANEL; kk++)
            {
                const int k = j0 + kk;
                if (sLcp[k] != sLcp[k + 1] && x[kk] != 0.0) // (myrow > k: these rows are at least 64 below the panel's first)
                {
                    const double l = x[kk] / clamp_pivot(Rb[kk * GETRF_PANEL + kk]);
                    x[kk] = l;
#pragma unroll
                    for (int c = 0; c < GETRF_PANEL; c++)
                        if (c > kk)
                            x[c] = x[c] - l * Rb[kk * GETRF_PANEL + c];
                }
            }
        }
        if (row_thread)
        {
#pragma unroll
            for (int c = 0; c < GETRF_PANEL; c++)
            {
                D[(size_t)(j0 + c) * nb + myrow] = x[c];
                P[c * ldp + myrow] = x[c];
            }
        }
        sync();
    };
    // ---- strip of panel j0 into image `buf`: thread t < nb - jt owns column jt + t of the 16 strip rows; forward
    // substitution with the unit lower 16 x 16 tile L11 read (broadcast) from the panel image
    auto strip = [&](int j0, int buf, auto sync)
    {
        const double *P = Pb + (size_t)buf * GETRF_PANEL * ldp;
        double *S = Sb + (size_t)buf * GETRF_PANEL * ldp;
        const int jt = j0 + GETRF_PANEL;
        if (tid < nb - jt)
        {
            const int c = jt + tid;
            double s[GETRF_PANEL];
            const double2 *src = reinterpret_cast<const double2 *>(D + (size_t)c * nb + j0);
#pragma unroll
            for (int q = 0; q < GETRF_PANEL / 2; q++)
            {
                const double2 v = src[q];
                s[2 * q] = v.x;
                s[2 * q + 1] = v.y;
            }
#pragma unroll
            for (int kk = 0; kk < GETRF_PANEL; kk++)
            {
                if (s[kk] != 0.0)
                {
#pragma unroll
                    for (int rr = 0; rr < GETRF_PANEL; rr++)
                        if (rr > kk)
                            s[rr] = s[rr] - P[kk * ldp + j0 + rr] * s[kk];
                }
            }
            double2 *dst = reinterpret_cast<double2 *>(D + (size_t)c * nb + j0);
#pragma unroll
            for (int q = 0; q < GETRF_PANEL / 2; q++)
                dst[q] = make_double2(s[2 * q], s[2 * q + 1]);
#pragma unroll
            for (int kk = 0; kk < GETRF_PANEL; kk++)
                S[kk * ldp + c] = s[kk];
        }
        sync();
    };
    // ---- one 16 x 16 tile (rows r0.., columns c0..) of the trailing block minus the rank-16 product of image `buf`
    // (A[i = l15][k = l4] = -U(k, c + i), B[k = l4][j = l15] = L(r + j, k); accumulator register g of lane l is
    // D(r + l15, c + l4 + 4g)): four tiles per pass, their loads together, then the MFMAs, then the stores
    auto update_tiles4 = [&](const int (&r0)[4], const int (&c0)[4], int ntile, int buf)
    {
        const double *P = Pb + (size_t)buf * GETRF_PANEL * ldp;
        const double *S = Sb + (size_t)buf * GETRF_PANEL * ldp;
        v4f64 t[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            t[u] = (v4f64){0.0, 0.0, 0.0, 0.0};
            if (u < ntile)
            {
#pragma unroll
                for (int g = 0; g < 4; g++)
                    t[u][g] = D[(size_t)(c0[u] + l4 + 4 * g) * nb + r0[u] + l15];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (u < ntile)
            {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    t[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(-S[(q * 4 + l4) * ldp + c0[u] + l15], P[(q * 4 + l4) * ldp + r0[u] + l15], t[u], 0, 0, 0);
            }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (u < ntile)
            {
#pragma unroll
                for (int g = 0; g < 4; g++)
                    D[(size_t)(c0[u] + l4 + 4 * g) * nb + r0[u] + l15] = t[u][g];
            }
    };
    auto wg_sync = []()
    { __syncthreads(); };

    // panel 0 and its strip by everyone, as in the kernel without look-ahead
    __syncthreads(); // (la_count)
    panel(0, 0, wg_sync);
    strip(0, 0, wg_sync);
    for (int j0 = 0; j0 + GETRF_PANEL < nb; j0 += GETRF_PANEL)
    {
        const int cur = (j0 / GETRF_PANEL) & 1, nxt = cur ^ 1;
        const int jt = j0 + GETRF_PANEL; // first trailing row / column
        const int mt = (nb - jt) / 16;   // trailing tiles per dimension (>= 1)
        if (wave < LA)
        {
            // (a) the tiles the next panel and the next strip consist of: tile column 0 (mt tiles) and the rest of tile
            // row 0 (mt - 1 tiles) of the trailing block, dealt over the look-ahead wavefronts four at a time
            const int npri = 2 * mt - 1;
            for (int base = wave * 4; base < npri; base += LA * 4)
            {
                int r0[4], c0[4], n = 0;
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    const int i = base + u;
                    r0[u] = c0[u] = jt;
                    if (i < npri)
                    {
                        r0[u] = i < mt ? jt + 16 * i : jt;            // column 0: rows i
                        c0[u] = i < mt ? jt : jt + 16 * (i - mt + 1); // row 0: columns 1..
                        n = u + 1;
                    }
                }
                update_tiles4(r0, c0, n, cur);
            }
            __builtin_amdgcn_s_waitcnt(0); // (stores of the tiles before the arrival: see la_barrier)
            la_barrier();
            // (b) next panel, (c) next strip, into the other images
            panel(jt, nxt, la_barrier);
            if (jt + GETRF_PANEL < nb)
                strip(jt, nxt, la_barrier);
        }
        else
        {
            // the rest of the trailing block (tile rows and columns >= 1) on the other wavefronts, 32 x 32 macro tiles
            const double *P = Pb + (size_t)cur * GETRF_PANEL * ldp;
            const double *S = Sb + (size_t)cur * GETRF_PANEL * ldp;
            const int m1 = mt - 1, mm = (m1 + 1) / 2;
            for (int mtile = wave - LA; mtile < mm * mm; mtile += nwaves - LA)
            {
                const int ri = (mtile % mm) * 2, ci = (mtile / mm) * 2;
                const int r0 = jt + 16 + ri * 16, c0 = jt + 16 + ci * 16;
                const bool hr = ri + 1 < m1, hc = ci + 1 < m1; // second row / column of tiles exists
                const int r1 = hr ? r0 + 16 : r0, c1 = hc ? c0 + 16 : c0;
                double a0[4], a1[4], b0[4], b1[4];
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    a0[q] = -S[(q * 4 + l4) * ldp + c0 + l15];
                    a1[q] = -S[(q * 4 + l4) * ldp + c1 + l15];
                    b0[q] = P[(q * 4 + l4) * ldp + r0 + l15];
                    b1[q] = P[(q * 4 + l4) * ldp + r1 + l15];
                }
                const bool za0 = !__any((a0[0] != 0.0) | (a0[1] != 0.0) | (a0[2] != 0.0) | (a0[3] != 0.0));
                const bool za1 = !hc || !__any((a1[0] != 0.0) | (a1[1] != 0.0) | (a1[2] != 0.0) | (a1[3] != 0.0));
                const bool zb0 = !__any((b0[0] != 0.0) | (b0[1] != 0.0) | (b0[2] != 0.0) | (b0[3] != 0.0));
                const bool zb1 = !hr || !__any((b1[0] != 0.0) | (b1[1] != 0.0) | (b1[2] != 0.0) | (b1[3] != 0.0));
                const bool d00 = !(zb0 || za0), d10 = !(zb1 || za0), d01 = !(zb0 || za1), d11 = !(zb1 || za1);
                v4f64 t00 = {0.0, 0.0, 0.0, 0.0}, t10 = t00, t01 = t00, t11 = t00;
#pragma unroll
                for (int g = 0; g < 4; g++)
                {
                    if (d00)
                        t00[g] = D[(size_t)(c0 + l4 + 4 * g) * nb + r0 + l15];
                    if (d10)
                        t10[g] = D[(size_t)(c0 + l4 + 4 * g) * nb + r1 + l15];
                    if (d01)
                        t01[g] = D[(size_t)(c1 + l4 + 4 * g) * nb + r0 + l15];
                    if (d11)
                        t11[g] = D[(size_t)(c1 + l4 + 4 * g) * nb + r1 + l15];
                }
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    if (d00)
                        t00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b0[q], t00, 0, 0, 0);
                    if (d10)
                        t10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b1[q], t10, 0, 0, 0);
                    if (d01)
                        t01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b0[q], t01, 0, 0, 0);
                    if (d11)
                        t11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b1[q], t11, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; g++)
                {
                    if (d00)
                        D[(size_t)(c0 + l4 + 4 * g) * nb + r0 + l15] = t00[g];
                    if (d10)
                        D[(size_t)(c0 + l4 + 4 * g) * nb + r1 + l15] = t10[g];
                    if (d01)
                        D[(size_t)(c1 + l4 + 4 * g) * nb + r0 + l15] = t01[g];
                    if (d11)
                        D[(size_t)(c1 + l4 + 4 * g) * nb + r1 + l15] = t11[g];
                }
            }
        }
        __syncthreads(); // the trailing block is up to date, the next panel and strip images are complete
    }

    if (T.defer_gather)
    {
        // the factors stay in the dense image; the diagonal tiles are saved behind the mirror (values + occupancy map)
        // because diag_tile_inverse_kernel replaces them by their inverses before the sparsify job reads the image
        double *__restrict__ saved = D + (size_t)nb * nb + MIRROR_MAP_BYTES / sizeof(double);
        for (int i = tid; i < nb * 16; i += 1024)
        {
            const int pt = i >> 8, cc = (i >> 4) & 15, rr = i & 15;
            saved[i] = D[(size_t)(16 * pt + cc) * nb + 16 * pt + rr];
        }
    }
    else
    {
        // gather the factors back into the sparse record: four entries per thread and pass, so that the index loads, the
        // searches and the reads of D of different entries overlap (one entry at a time is a chain of three dependent L2
        // round trips per entry: 56 of the kernel's 330 us)
        constexpr int GU = 4;
        for (u32 p0 = tid; p0 < nnzL; p0 += GU * 1024)
        {
            u32 r[GU];
            double v[GU];
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * 1024;
                r[u] = p < nnzL ? T.lri[p] : 0u;
            }
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * 1024;
                v[u] = p < nnzL ? D[(size_t)owner_of(sLcp, nb, p) * nb + r[u]] : 0.0;
            }
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * 1024;
                if (p < nnzL)
                    T.lval[p] = v[u];
            }
        }
        for (u32 p0 = tid; p0 < nnzU; p0 += GU * 1024)
        {
            u32 c[GU];
            double v[GU];
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * 1024;
                c[u] = p < nnzU ? T.uci[p] : 0u;
            }
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * 1024;
                v[u] = p < nnzU ? D[(size_t)c[u] * nb + owner_of(sUrp, nb, p)] : 0.0;
            }
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * 1024;
                if (p < nnzU)
                    T.uval[p] = v[u];
            }
        }
    }
    __syncthreads();
    GETRF_STAMP(6)
    if (T.invert_tiles)
    {
        // one wavefront per diagonal tile, 16 x 17 doubles of the (now free) panel image each
        double(*Tw)[17] = reinterpret_cast<double(*)[17]>(smem_raw) + wave * 16;
        for (int p0 = 0; p0 < nb / 16; p0 += nwaves)
            invert_diag_tile(D, nb, p0 + wave, Tw, lane, p0 + wave < nb / 16, []()
                             { __syncthreads(); });
    }
    ops = wave_sum(ops);
    if (lane == 0 && ops)
        atomicAdd(flop_counter, ops);
}
#endif // PG_DENSE_PANELS
