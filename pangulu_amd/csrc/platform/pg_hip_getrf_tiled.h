// pg_hip_getrf_tiled.h -- GETRF of one dense-mode diagonal block per workgroup, organised around 16 x 16 tiles with
// STATIC tile ownership and a dedicated factorisation wavefront.  (included by pg_hip_platform.hip; R64, nb % 16 == 0,
// nb <= 256.)  Replaces densify + cuSOLVER getrf + gather of the reference's GPU path (…0201000.cu:547-641); the
// arithmetic is the CPU kernel's right-looking elimination without pivoting (…0100000.c:57-135) in a blocked order.
//
// Why: the earlier blocked kernels (getrf_blocked / getrf_lookahead, tools/experiments/pg_hip_getrf_blocked.h since round 6) spend ~300-360 us on a 256 x 256 block
// whatever its fill: every panel step is a chain of L2 round trips (panel load, strip load, trailing tiles written by
// other wavefronts and read back), 16 + 16 pivot steps with an IEEE division each, barriers in between, and an epilogue
// that inverts the diagonal tiles.  Near the root of the elimination tree that chain IS the factorisation's critical path.
//
// Here, per panel step `it` (panel `it` is being factorised while panel it-1 is applied to the trailing block):
//   * seven COMPUTE wavefronts own the 16 x 16 tiles of the block cyclically along both dimensions: tile (ti, tj)
//     belongs to wavefront (ti + 3 tj) mod 7, so every tile row and every tile column is spread over all seven.  (Eight
//     wavefronts per workgroup: two per SIMD get 256 registers each, enough to keep ten trailing tiles in flight per
//     pass.)  A tile is only ever loaded / stored by its owner, so the trailing block needs no synchronisation through
//     memory at all: every wavefront applies panel it-1 to its own tiles on the matrix cores (operands from the LDS
//     images of the panel and the strip) and writes them back.
//   * tiles of tile column `it` / tile row `it` (the next panel / strip) are updated FIRST and stay in their owner's
//     registers; the diagonal tile (it, it) is kept in its owner's registers from the step before (no memory round trip
//     on the critical path) and goes to LDS right at the start of the step.
//   * wavefront 7 factorises that diagonal tile (rows in lanes, pivot rows broadcast with v_readlane, pivot
//     reciprocals by v_rcp_f64 + two Newton steps + one correction of the quotient instead of an IEEE division per
//     row) while the others work on the trailing block, and publishes L11\U11 + the reciprocal pivots in LDS.
//   * the rest of the panel (rows below the tile: X = T U11^-1) and of the strip (Y = L11^-1 T) are finished by
//     substitution in the LDS images, one row / column per thread, U11 / L11 read as LDS broadcasts; on gfx950 the
//     vector FMA rate equals the f64 matrix-core rate, so substitution (half the flops of a product with an explicit
//     inverse, and no inverse on the critical path) is the cheaper form.  ONE workgroup barrier per step.
//   * the dense TSTRF/GESSM of this level want the INVERSES of the diagonal tiles in the image (pg_hip_trsm_dense.h):
//     one pass at the end, two tiles per wavefront, a quarter wavefront per triangular factor.
//   * structurally empty tiles are skipped by the block's 16 x 16-bit occupancy map (from the mirror when the block
//     arrives as a dense mirror, built while scattering otherwise): tile (ti, tj) takes part in step k only if tiles
//     (ti, k) and (k, tj) hold pattern entries.
// Order of operations per entry: updates still arrive in ascending pivot order; inside one 16-wide panel they are
// summed by the matrix cores (as before).  Multiplying by a refined reciprocal instead of dividing moves L entries by at
// most one unit in the last place.  Parity: within 1e-12 of the oracle
// (tests/test_gpu_parity*.py); the order-preserving kernel (GETRF_STRICT_ORDER) is unchanged.
#pragma once
// (shared with tools/experiments/pg_hip_getrf_blocked.h, where these lived until round 6)
#ifndef GETRF_STAMP
// index i with ptr[i] <= p < ptr[i+1] (ptr ascending, ptr[0] = 0, p < ptr[n])
__device__ inline int owner_of(const u32 *ptr, int n, u32 p)
{
    int lo = 0, hi = n; // invariant: ptr[lo] <= p < ptr[hi]
    while (hi - lo > 1)
    {
        const int mid = (lo + hi) >> 1;
        if (ptr[mid] <= p)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

#define GETRF_STAMP(slot)                                                  \
    if (dbg && tid == 0 && blockIdx.x == 0)                                \
    {                                                                      \
        unsigned long long now_ = __builtin_amdgcn_s_memtime();            \
        dbg[slot] += now_ - stamp_;                                        \
        stamp_ = now_;                                                     \
    }
#endif


#define GT_THREADS 512
#define GT_COMPUTE_WAVES 7
#define GT_CHUNK 10 // tiles of the trailing block a wavefront keeps in registers per pass (8 registers each)

// dynamic LDS of getrf_tiled_f64_kernel (the kernel lays its arrays out in this order)
__host__ __device__ inline size_t gt_lds_bytes(int nb)
{
    return sizeof(double) * (4 * 16 * (size_t)(nb + 2) + 16 * 17 + 16 + 256) // panel / strip images, Td, rdiag, rd_all
           + sizeof(unsigned) * (2 * (size_t)(nb + 1) + 16 + 4)           // pointer arrays, occupancy map, flags
           + 64 * GT_COMPUTE_WAVES                                        // tile lists
           + sizeof(double) * 3 * 16 * 17;                                // chase variant: factorised tile + its two inversion images
}

__device__ inline double gt_refined_rcp(double p)
{
    double r = __builtin_amdgcn_rcp(p);
    double e = __builtin_fma(-p, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-p, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}

__device__ inline double gt_readlane(double v, int lane)
{
    union
    {
        double d;
        int w[2];
    } x;
    x.d = v;
    x.w[0] = __builtin_amdgcn_readlane(x.w[0], lane);
    x.w[1] = __builtin_amdgcn_readlane(x.w[1], lane);
    return x.d;
}

// stores of what another workgroup reads WHILE this kernel runs (CHASE: the dense solves of the level start on panel p as soon
// as this factorisation has published it -- pg_hip_platform.hip, getrf_trsm_chase_kernel): `sc1`, the producer half of the
// hand-off MI355X_MICROARCH.md tabulates (sc1 stores, every storing wave waits vmcnt(0), one lane stores the flag sc1 behind a
// workgroup barrier; the consumer polls the flag and loads the bytes sc1)
template <bool CHASE>
__device__ __forceinline__ void gt_publish(double *p, double v)
{
    if (CHASE)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}

template <bool CHASE>
__device__ __forceinline__ void getrf_tiled_body(const GetrfTaskD &T, int nb, unsigned long long *flop_counter, unsigned long long *dbg, unsigned *progress)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int ldp = nb + 2;
    double *Pb = reinterpret_cast<double *>(smem_raw); // Pb[(buf * 16 + c) * ldp + r]: column c of a panel, row r (absolute)
    double *Sb = Pb + 2 * 16 * ldp;                    // Sb[(buf * 16 + k) * ldp + c]: row k of a strip, column c (absolute)
    double(*Td)[17] = reinterpret_cast<double(*)[17]>(Sb + 2 * 16 * ldp); // diagonal tile: pre-LU, then L11\U11
    double *rdiag = reinterpret_cast<double *>(Td) + 16 * 17;              // reciprocals of the (clamped) pivots of the tile
    double *rd_all = rdiag + 16;                                           // rd_all[it * 16 + k]: the same, kept for the epilogue
    u32 *sLcp = reinterpret_cast<u32 *>(rd_all + 256);
    u32 *sUrp = sLcp + nb + 1;
    unsigned *smap = sUrp + nb + 1; // smap[tj] bit ti: tile (ti, tj) holds pattern entries
    unsigned *flags = smap + 16;    // [0] diagonal tile published, [1] tile factorised, [2] priority tiles in the images
    unsigned char *tlist = reinterpret_cast<unsigned char *>(flags + 4); // per compute wavefront: its tiles of the current step (ti << 4 | tj)
    double *Tk = reinterpret_cast<double *>(tlist + 64 * GT_COMPUTE_WAVES); // CHASE: the tile factorised last, [row][17]; then [plain, reversed][16][17]
    double *__restrict__ D = reinterpret_cast<double *>(T.dense);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int nt = nb / 16;
    unsigned long long stamp_ = dbg ? __builtin_amdgcn_s_memtime() : 0;

    // ---- prologue: dense image + occupancy map -----------------------------------------------------------------
    if (tid < 16)
        smap[tid] = 0;
    if (tid < 4)
        flags[tid] = 0;
    if (!T.preloaded)
    {
        for (int i = tid; i < nb * nb / 2; i += GT_THREADS)
            reinterpret_cast<double2 *>(D)[i] = make_double2(0.0, 0.0);
    }
    for (int i = tid; i <= nb; i += GT_THREADS)
    {
        sLcp[i] = T.lcp[i];
        sUrp[i] = T.urp[i];
    }
    __syncthreads();
    const u32 nnzL = sLcp[nb], nnzU = sUrp[nb];
    unsigned long long ops = 0;
    if (!T.preloaded)
    {
        for (u32 p = tid; p < nnzL; p += GT_THREADS)
        {
            const int c = owner_of(sLcp, nb, p);
            const u32 r = T.lri[p];
            D[(size_t)c * nb + r] = T.lval[p];
            atomicOr(&smap[c >> 4], 1u << (r >> 4));
        }
        for (u32 p = tid; p < nnzU; p += GT_THREADS)
        {
            const int r = owner_of(sUrp, nb, p);
            const u32 c = T.uci[p];
            D[(size_t)c * nb + r] = T.uval[p];
            atomicOr(&smap[c >> 4], 1u << (r >> 4));
        }
    }
    else if (tid < 16)
        smap[tid] = mirror_map(D, nb)[tid];
    for (int c = tid; c < nb; c += GT_THREADS)
    {
        // structural flop count of the sparse algorithm (what the reference counts, src/pangulu_kernel_interface.c:4-82)
        const u32 nl = sLcp[c + 1] - sLcp[c], nu = sUrp[c + 1] - sUrp[c];
        if (nu > 0)
            ops += (unsigned long long)nl * (1ull + 2ull * (nu - 1));
    }
    __syncthreads();
    // the image is a mirror: the dense solves of this level skip structurally empty factor tiles by this map
    if (CHASE)
    {
        if (T.invert_tiles && tid < 8)
            __hip_atomic_store(reinterpret_cast<unsigned *>(D + (size_t)nb * nb) + tid, (smap[2 * tid] & 0xFFFFu) | (smap[2 * tid + 1] << 16), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    else if (T.invert_tiles && tid < 16)
        reinterpret_cast<unsigned short *>(D + (size_t)nb * nb)[tid] = (unsigned short)smap[tid];
    GETRF_STAMP(0)

    auto live = [&](int ti, int tj) -> bool
    { return (smap[tj] >> ti) & 1u; };
    auto wait_flag = [&](int f, unsigned target)
    {
        unsigned spins = 0;
        while (__hip_atomic_load(&flags[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target)
        {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24)) // (seconds: a lost flag must abort the launch, not hang the device)
                __builtin_trap();
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    auto post_flag = [&](int f, bool add, unsigned v)
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0)
        {
            if (add)
                __hip_atomic_fetch_add(&flags[f], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else
                __hip_atomic_store(&flags[f], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };

    auto owner = [&](int ti, int tj) -> int
    { return (ti + 3 * tj) % 7; }; // compute wavefront that owns tile (ti, tj)
    v4f64 dnext = {0.0, 0.0, 0.0, 0.0};     // diagonal tile of the NEXT step when this wavefront owns it (column form)
    bool have_dnext = false;
    double *saved = D + (size_t)nb * nb + MIRROR_MAP_BYTES / sizeof(double);
    // 32-bit byte offsets into the image (nb <= 256: below 512 KiB): one uniform base + one VGPR per access
    const unsigned colB = (unsigned)nb * 8u, col4B = 4u * colB;
    const unsigned cf = ((unsigned)l4 * (unsigned)nb + (unsigned)l15) * 8u; // column form: register g of lane l is (r0 + l15, c0 + l4 + 4g)
    const unsigned rf = ((unsigned)l15 * (unsigned)nb + (unsigned)l4) * 8u; // row form:    register g of lane l is (r0 + l4 + 4g, c0 + l15)
    auto gd = [&](unsigned byteoff) -> double &
    { return *reinterpret_cast<double *>(reinterpret_cast<char *>(D) + byteoff); };
    // inverse of one triangular factor of diagonal tile pt by a quarter wavefront (lane groups 0 / 2: U via the index-reversed
    // image M2[1], groups 1 / 3: L via the plain image M2[0]), written over the tile in the image
    auto invert_quarter = [&](const double *M2, int pt)
    {
        const int grp = (lane >> 4) & 1, c = l15;
        const double *M = M2 + (grp ? 0 : 1) * 272;
        const int cc = grp ? c : 15 - c;
        double z[16];
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            double s0 = (r == cc) ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
            for (int k = 0; k < r; k++)
            {
                if (k & 1)
                    s1 = __builtin_fma(-M[r * 17 + k], z[k], s1);
                else
                    s0 = __builtin_fma(-M[r * 17 + k], z[k], s0);
            }
            z[r] = grp ? s0 + s1 : (s0 + s1) * rd_all[pt * 16 + 15 - r];
        }
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            const int row = grp ? r : 15 - r;
            if (grp ? (row > c) : (row <= c))
                gt_publish<CHASE>(&D[(size_t)(16 * pt + c) * nb + 16 * pt + row], z[r]);
        }
    };
    // CHASE, factorisation wavefront: the tile it factorised last (kept in Tk) -> both inversion images -> inverses into the image
    auto invert_kept_tile = [&](int pt)
    {
        double v4[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
            v4[u] = Tk[((lane + 64 * u) & 15) * 17 + ((lane + 64 * u) >> 4)]; // entry (rr, cc) of the tile, e = cc << 4 | rr
        wave_lds_fence();
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            const int e = lane + 64 * u, cc = e >> 4, rr = e & 15;
            Tk[272 + rr * 17 + cc] = v4[u];
            Tk[2 * 272 + (15 - rr) * 17 + (15 - cc)] = v4[u];
        }
        wave_lds_fence();
        if (lane < 32)
            invert_quarter(Tk + 272, pt);
    };

    // phase stamps (PANGULU_HIP_DEBUG_GETRF): block 0 only; slots 1..4 by compute wavefront 0, slot 7 by the factorisation wavefront
    unsigned long long ph_ = 0;
#define GT_PHASE(slot)                                                    \
    if (dbg && lane == 0 && blockIdx.x == 0)                              \
    {                                                                     \
        unsigned long long now_ = __builtin_amdgcn_s_memtime();           \
        dbg[slot] += now_ - ph_;                                          \
        ph_ = now_;                                                       \
    }
    for (int it = 0; it < nt; it++)
    {
        if (dbg)
            ph_ = __builtin_amdgcn_s_memtime();
        // panel it-1 (image `cur`) is applied; panel `it` is built into image `nxt`
        const int cur = (it + 1) & 1, nxt = it & 1;
        const double *P = Pb + (size_t)cur * 16 * ldp, *S = Sb + (size_t)cur * 16 * ldp;
        double *Pn = Pb + (size_t)nxt * 16 * ldp, *Sn = Sb + (size_t)nxt * 16 * ldp;
        const int k0 = it * 16;        // first row / column of panel `it`
        const int kp = (it - 1) * 16;  // of the panel being applied
        const bool apply = it > 0;
        const int tp = it - 1;

        if (wave == GT_COMPUTE_WAVES)
        {
            // ---- factorisation wavefront: LU of the diagonal tile (it, it) ------------------------------------------
            if (CHASE && it > 0 && T.invert_tiles)
                invert_kept_tile(it - 1); // (while the owner of tile (it, it) is still updating it)
            wait_flag(0, (unsigned)it + 1);
            if (dbg)
                ph_ = __builtin_amdgcn_s_memtime();
            double x[16];
#pragma unroll
            for (int c = 0; c < 16; c++)
                x[c] = Td[l15][c]; // (lanes 16.. repeat rows 0..15; only lanes 0..15 write back)
#pragma unroll
            for (int k = 0; k < 16; k++)
            {
                double u[16];
#pragma unroll
                for (int c = 0; c < 16; c++)
                    if (c >= k)
                        u[c] = gt_readlane(x[c], k);
                double p = u[k];
                if ((p < 0 ? -p : p) < PANGULU_TOL)
                    p = PANGULU_TOL;
                const double rp = gt_refined_rcp(p);
                if (lane == k)
                {
                    rdiag[k] = rp;
                    rd_all[it * 16 + k] = rp;
                }
                if (l15 > k)
                {
                    double l = x[k] * rp;
                    l = __builtin_fma(__builtin_fma(-l, p, x[k]), rp, l); // one correction: the quotient to the last place
                    x[k] = l;
#pragma unroll
                    for (int c = 0; c < 16; c++)
                        if (c > k)
                            x[c] = __builtin_fma(-l, u[c], x[c]);
                }
            }
            if (lane < 16)
            {
#pragma unroll
                for (int c = 0; c < 16; c++)
                    Td[lane][c] = x[c];
                if (CHASE)
                {
#pragma unroll
                    for (int c = 0; c < 16; c++)
                        Tk[lane * 17 + c] = x[c];
                }
            }
            post_flag(1, false, (unsigned)it + 1);
            GT_PHASE(7)
            if (lane < 16 && T.defer_gather)
            {
                // the factorised tile goes behind the mirror for the deferred sparsify job
#pragma unroll
                for (int c = 0; c < 16; c++)
                    saved[(it << 8) + (c << 4) + lane] = x[c];
            }
            if (lane < 16 && !(T.invert_tiles && T.defer_gather))
            {
#pragma unroll
                for (int c = 0; c < 16; c++)
                    gd((unsigned)((k0 + c) * nb + k0 + lane) * 8u) = x[c];
            }
        }
        else
        {
            // ---- compute wavefronts ---------------------------------------------------------------------------------
            // (a) the diagonal tile of this step, by its owner: from registers (or memory in the first two steps)
            if (owner(it, it) == wave)
            {
                v4f64 t;
                if (have_dnext)
                    t = dnext;
                else
                {
#pragma unroll
                    for (int g = 0; g < 4; g++)
                        t[g] = gd(cf + (unsigned)(k0 * nb + k0) * 8u + g * col4B);
                }
                have_dnext = false;
                if (apply && live(it, tp) && live(tp, it))
                {
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        t = __builtin_amdgcn_mfma_f64_16x16x4f64(-S[(q * 4 + l4) * ldp + k0 + l15], P[(q * 4 + l4) * ldp + k0 + l15], t, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; g++)
                    Td[l15][l4 + 4 * g] = t[g];
                post_flag(0, false, (unsigned)it + 1);
            }
            // (b) priority tiles: tile column `it` (rows below the diagonal tile: the next panel) and tile row `it` (columns
            //     right of it: the next strip), updated by panel it-1 first and then parked in the LDS images of the next panel /
            //     strip until the diagonal tile's inverses are there.  A wavefront owns at most three of each; all their
            //     loads go out together
            const int ti0 = ((wave - 3 * it) % 7 + 7) % 7;   // rows ti = ti0 (mod 7) of tile column `it` are this wavefront's
            const int tj0 = (((wave - it) % 7 + 7) * 5) % 7; // columns tj = tj0 (mod 7) of tile row `it`
            bool okc[3], okr[3];
            v4f64 tc[3], tr[3];
#pragma unroll
            for (int u = 0; u < 3; u++)
            {
                const int ti = ti0 + 7 * u, tj = tj0 + 7 * u;
                okc[u] = ti > it && ti < nt && live(ti, it);
                okr[u] = tj > it && tj < nt && live(it, tj);
                if (okc[u])
                {
#pragma unroll
                    for (int g = 0; g < 4; g++)
                        tc[u][g] = gd(cf + (unsigned)(k0 * nb + ti * 16) * 8u + g * col4B);
                }
                if (okr[u])
                {
#pragma unroll
                    for (int g = 0; g < 4; g++) // row form: register g of lane l is D(k0 + l4 + 4g, c0 + l15)
                        tr[u][g] = gd(rf + (unsigned)(tj * 16 * nb + k0) * 8u + 32u * g);
                }
            }
            if (apply)
            {
#pragma unroll
                for (int u = 0; u < 3; u++)
                {
                    const int ti = ti0 + 7 * u, tj = tj0 + 7 * u;
                    if (okc[u] && live(ti, tp) && live(tp, it))
                    {
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            tc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(-S[(q * 4 + l4) * ldp + k0 + l15], P[(q * 4 + l4) * ldp + ti * 16 + l15], tc[u], 0, 0, 0);
                    }
                    if (okr[u] && live(it, tp) && live(tp, tj))
                    {
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            tr[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(-P[(q * 4 + l4) * ldp + k0 + l15], S[(q * 4 + l4) * ldp + tj * 16 + l15], tr[u], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 3; u++)
            {
                if (okc[u])
                {
#pragma unroll
                    for (int g = 0; g < 4; g++)
                        Pn[(l4 + 4 * g) * ldp + (ti0 + 7 * u) * 16 + l15] = tc[u][g];
                }
                if (okr[u])
                {
#pragma unroll
                    for (int g = 0; g < 4; g++)
                        Sn[(l4 + 4 * g) * ldp + (tj0 + 7 * u) * 16 + l15] = tr[u][g];
                }
            }
            post_flag(2, true, 1u);
            if (wave == 0)
            {
                GT_PHASE(1)
            }
            __builtin_amdgcn_sched_barrier(0); // (keeps the loads of (c) from being hoisted into (b): registers)
            // (c) the rest of the trailing block: own tiles with ti, tj > it that panel it-1 reaches, applied in memory.  Every
            //     pass is a round trip to L2 (load, matrix cores, store; on gfx9-family ISAs loads and stores share one counter,
            //     so a pass cannot overlap the next one's loads with its own stores): the wavefront lists its tiles of this step
            //     and takes up to GT_CHUNK of them per pass -- two passes in the first steps, one afterwards
            if (apply)
            {
                unsigned char *mylist = tlist + wave * 64;
                int n = 0;
                for (int base = 0; base < 256; base += 64)
                {
                    const int idx = base + lane, ti = idx >> 4, tj = idx & 15;
                    const bool mine = ti > it && tj > it && ti < nt && tj < nt && (ti + 3 * tj) % 7 == wave && live(ti, tp) && live(tp, tj) && live(ti, tj);
                    const unsigned long long m = __ballot(mine);
                    if (mine)
                        mylist[n + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned char)idx;
                    n += __popcll(m);
                }
                wave_lds_fence();
                for (int base = 0; base < n; base += GT_CHUNK)
                {
                    v4f64 t[GT_CHUNK];
#pragma unroll
                    for (int u = 0; u < GT_CHUNK; u++)
                        if (base + u < n)
                        {
                            const int e = __builtin_amdgcn_readfirstlane((int)mylist[base + u]);
                            const unsigned o = cf + (unsigned)(((e & 15) * nb + (e >> 4)) * 16) * 8u;
#pragma unroll
                            for (int g = 0; g < 4; g++)
                                t[u][g] = gd(o + g * col4B);
                        }
#pragma unroll
                    for (int u = 0; u < GT_CHUNK; u++)
                        if (base + u < n)
                        {
                            const int e = __builtin_amdgcn_readfirstlane((int)mylist[base + u]);
                            const int r0 = (e >> 4) * 16, c0 = (e & 15) * 16;
#pragma unroll
                            for (int q = 0; q < 4; q++)
                                t[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(-S[(q * 4 + l4) * ldp + c0 + l15], P[(q * 4 + l4) * ldp + r0 + l15], t[u], 0, 0, 0);
                        }
#pragma unroll
                    for (int u = 0; u < GT_CHUNK; u++)
                        if (base + u < n)
                        {
                            const int e = __builtin_amdgcn_readfirstlane((int)mylist[base + u]);
                            if (e == (it + 1) * 17)
                            {
                                dnext = t[u]; // next step's diagonal tile stays in registers
                                have_dnext = true;
                            }
                            else
                            {
                                const unsigned o = cf + (unsigned)(((e & 15) * nb + (e >> 4)) * 16) * 8u;
#pragma unroll
                                for (int g = 0; g < 4; g++)
                                    gd(o + g * col4B) = t[u][g];
                            }
                        }
                }
            }
        }
        if (wave == 0)
        {
            GT_PHASE(2)
        }
        // (d) finish the panel and the strip by substitution in the LDS images, one row (wavefronts 0..3) / one column
        //     (wavefronts 4..7: the factorisation wavefront has nothing else on the critical path now) per thread, then write
        //     them to the image in memory
        wait_flag(2, (unsigned)GT_COMPUTE_WAVES * ((unsigned)it + 1));
        wait_flag(1, (unsigned)it + 1);
        if (wave == 0)
        {
            GT_PHASE(3)
        }
        if (wave < 4)
        {
            const int r = k0 + 16 + tid;
            if (r < nb && live(r >> 4, it))
            {
                double x[16];
#pragma unroll
                for (int c = 0; c < 16; c++)
                    x[c] = Pn[c * ldp + r];
#pragma unroll
                for (int k = 0; k < 16; k++)
                {
                    const double xk = x[k] * rdiag[k];
                    x[k] = xk;
#pragma unroll
                    for (int c = 0; c < 16; c++)
                        if (c > k)
                            x[c] = __builtin_fma(-xk, Td[k][c], x[c]);
                }
#pragma unroll
                for (int c = 0; c < 16; c++)
                {
                    Pn[c * ldp + r] = x[c];
                    gt_publish<CHASE>(&gd((unsigned)((k0 + c) * nb + r) * 8u), x[c]);
                }
            }
        }
        else
        {
            const int c = k0 + 16 + (tid - 256);
            if (c < nb && live(it, c >> 4))
            {
                double s[16];
#pragma unroll
                for (int k = 0; k < 16; k++)
                    s[k] = Sn[k * ldp + c];
#pragma unroll
                for (int kk = 0; kk < 16; kk++)
                {
#pragma unroll
                    for (int rr = 0; rr < 16; rr++)
                        if (rr > kk)
                            s[rr] = __builtin_fma(-Td[rr][kk], s[kk], s[rr]);
                }
                double2 *dst = reinterpret_cast<double2 *>(D + (size_t)c * nb + k0);
                if (CHASE)
                {
#pragma unroll
                    for (int k = 0; k < 16; k++)
                        gt_publish<true>(D + (size_t)c * nb + k0 + k, s[k]);
                }
                else
                {
#pragma unroll
                    for (int q = 0; q < 8; q++)
                        dst[q] = make_double2(s[2 * q], s[2 * q + 1]);
                }
#pragma unroll
                for (int k = 0; k < 16; k++)
                    Sn[k * ldp + c] = s[k];
            }
        }
        if (wave == 0)
        {
            GT_PHASE(4)
        }
        if (CHASE)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (every storing wave, before the word that announces the stores)
        __syncthreads(); // images of panel `it` complete; everyone is done with the images of panel it-1
        if (CHASE && tid == 0 && it > 0)
            __hip_atomic_store(progress, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // panels < it: factors and tile inverses are in memory
    }
    GETRF_STAMP(5)

    if (!T.defer_gather)
    {
        // gather the factors back into the sparse record (blocks without a mirror / without the records stream)
        constexpr int GU = 4;
        for (u32 p0 = tid; p0 < nnzL; p0 += GU * GT_THREADS)
        {
            u32 r[GU];
            double v[GU];
#pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * GT_THREADS;
                r[u] = p < nnzL ? T.lri[p] : 0u;
            }
#pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * GT_THREADS;
                v[u] = p < nnzL ? D[(size_t)owner_of(sLcp, nb, p) * nb + r[u]] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * GT_THREADS;
                if (p < nnzL)
                    T.lval[p] = v[u];
            }
        }
        for (u32 p0 = tid; p0 < nnzU; p0 += GU * GT_THREADS)
        {
            u32 c[GU];
            double v[GU];
#pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * GT_THREADS;
                c[u] = p < nnzU ? T.uci[p] : 0u;
            }
#pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * GT_THREADS;
                v[u] = p < nnzU ? D[(size_t)c[u] * nb + owner_of(sUrp, nb, p)] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * GT_THREADS;
                if (p < nnzU)
                    T.uval[p] = v[u];
            }
        }
    }
    __syncthreads();
    if (CHASE)
    {
        // the last tile's inverses, then the final word: every panel is there
        if (wave == GT_COMPUTE_WAVES && T.invert_tiles)
            invert_kept_tile(nt - 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0)
            __hip_atomic_store(progress, (unsigned)nt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    else if (T.invert_tiles)
    {
        // The dense TSTRF/GESSM of this level want the INVERSES of the diagonal tiles in the image (U11^-1 on and above
        // the diagonal, L11^-1 below: pg_hip_trsm_dense.h).  One pass: every wavefront takes two tiles, a quarter
        // wavefront per triangular factor -- lanes 0..15 / 32..47 solve U x = e_c on an index-reversed copy of their tile
        // (which makes it a forward substitution like the other), lanes 16..31 / 48..63 solve L y = e_c.
        double *Tw = reinterpret_cast<double *>(smem_raw) + (size_t)wave * (4 * 16 * 17); // [tile][plain, reversed][16][17]
        for (int i = lane; i < 512; i += 64)
        {
            const int t = i >> 8, e = i & 255, cc = e >> 4, rr = e & 15, pt = 2 * wave + t;
            if (pt < nt)
            {
                const double v = T.defer_gather ? saved[(pt << 8) + e] : D[(size_t)(16 * pt + cc) * nb + 16 * pt + rr];
                Tw[(t * 2 + 0) * 272 + rr * 17 + cc] = v;
                Tw[(t * 2 + 1) * 272 + (15 - rr) * 17 + (15 - cc)] = v;
            }
        }
        wave_lds_fence();
        const int t = lane >> 5, pt = 2 * wave + t;
        if (pt < nt)
            invert_quarter(Tw + (size_t)t * 2 * 272, pt);
    }
    GETRF_STAMP(6)
    ops = wave_sum(ops);
    if (lane == 0 && ops)
        atomicAdd(flop_counter, ops);
}


__global__ __launch_bounds__(GT_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void getrf_tiled_f64_kernel(
    const GetrfTaskD *__restrict__ tasks, int nb, unsigned long long *flop_counter, unsigned long long *dbg)
{
    const GetrfTaskD T = tasks[blockIdx.x];
    getrf_tiled_body<false>(T, nb, flop_counter, dbg, nullptr);
}
