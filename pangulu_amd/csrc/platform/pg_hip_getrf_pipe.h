// pg_hip_getrf_pipe.h -- GETRF of one dense-mode diagonal block per workgroup with the trailing block RESIDENT IN REGISTERS
// (round 5; included by pg_hip_platform.hip after pg_hip_getrf_tiled.h; R64 mirrors, nb = 128 or 256).
// Replaces densify + cuSOLVER getrf + gather of the reference's GPU path (...0201000.cu:547-641); the arithmetic is the CPU kernel's
// right-looking elimination without pivoting (...0100000.c:57-135) in a blocked order, as in the tiled kernel.
//
// Why another one.  The tiled kernel (pg_hip_getrf_tiled.h) takes 205 us for a lone dense 256 x 256 block where one CU's matrix
// cores need 36: every panel step writes its share of the trailing block back to L2 and reads it again (each pass a round trip of
// 1 us: loads and stores share one in-order counter), finishes the panel by 16-step substitutions (2.7 us) behind the trailing
// passes, and keeps 132 KB of LDS, i.e. a CU to itself -- which it does not get while an update launch is still handing out
// workgroups (0.7 ms beside one, 6.2 ms at worst; profiles/r04ao_elastic3d_77.md).  Near the root of the elimination tree that
// chain IS the factorisation (shell(398): 13.9 of 37 ms).
//
// Here (512 threads, 256 registers per wavefront):
//   * six TRAILING wavefronts own the 16 x 16 tiles 2D-cyclically (a 2 x 3 grid anchored at the bottom right corner) and keep the
//     tiles of the last 12 tile rows and columns -- 144 of 256 at nb = 256, all 64 at nb = 128 -- IN REGISTERS from the prologue
//     to the step that finishes them: 24 tiles = 192 registers per wavefront.  A step's update of a resident tile is four MFMAs on
//     operands from the LDS images of the panel and the strip; nothing of the trailing block travels.
//   * the band of the first tile rows / columns that does not fit (tiles with min(i, j) < 4 at nb = 256) is handled LEFT-LOOKING: a
//     band tile stays in memory untouched until the step that finishes it, and then receives all its (at most three) updates at
//     once from the finished factor tiles in memory (same CU: workgroup-scope visibility, a drained store counter and a barrier).
//   * panel and strip tiles are finished by their OWNERS on the matrix cores with the inverses of the diagonal tile's factors
//     (X = T U11^-1, Y = L11^-1 T; the accumulator layout of a tile is the operand layout of the second MFMA source, so a resident
//     tile is an operand as it stands); those inverses are what the dense TSTRF/GESSM of the level want in the image anyway
//     (pg_hip_trsm_dense.h), so nothing is computed twice, and the 16-step substitutions are gone.
//   * ONE wavefront factorises the diagonal tile and inverts its two factors (a quarter wavefront each) while the others apply
//     the previous panel to the trailing block; three workgroup barriers per step:
//         A  diagonal tile k in LDS      | wavefront 0: LU + inverses        ||  trailing wavefronts: panel k-1 on their resident tiles
//         B  inverses in LDS, images free| owners finish panel / strip tiles of step k into the images and the block's image in memory
//         C  images of step k complete   | the owner of diagonal tile k+1 applies panel k to it and hands it over
//   * 69 KB of LDS (one image pair: a step's tiles are finished behind barrier B, when nobody reads the previous ones any more):
//     two factorisations, or one beside an update workgroup, share a CU.
// Order of operations per entry: updates in ascending pivot order panel by panel, summed by the matrix cores inside a panel (as the
// tiled kernel); the panel solves multiply by explicit 16 x 16 inverses instead of substituting (as the dense TSTRF/GESSM do).
// Parity: within 1e-12 of the oracle (tests/test_gpu_parity*.py run every case on this kernel by default; PANGULU_HIP_GETRF_PIPE=0
// selects the tiled kernel, which also still serves images that are not mirrors or whose sparse record is gathered in the kernel).
#pragma once

#define GP_THREADS 512
#define GP_TWAVES 6

__host__ __device__ inline size_t gp_lds_bytes(int nb)
{
    return sizeof(double) * (2 * 16 * (size_t)(nb + 2) + 16 * 17 /* Td */ + 2 * 16 * 17 /* inversion images */ + 2 * 16 * 17 /* IL, IU */ + 48 /* rdiag, row scalings of the inversion */) +
           sizeof(unsigned) * (2 * (size_t)(nb + 1) + 16);
}

__device__ __forceinline__ void gp_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// The factorisation wavefront of getrf_pipe_f64_kernel: LU of diagonal tile k (rows in lanes, pivot rows by v_readlane, refined
// reciprocals), then the inverses of its two factors (a quarter wavefront each), for k = 0 .. NT - 1, in step with the trailing
// wavefronts through the workgroup's three barriers per step.
template <int NT>
__device__ __forceinline__ void gp_factor_wavefront(double *__restrict__ D, double (*Td)[17], double *Mi, double (*IL)[17], double (*IU)[17], double *rdiag,
                                                               unsigned long long *dbg)
{
    constexpr int nb = NT * 16;
    constexpr unsigned colB = (unsigned)nb * 8u;
    const int lane = threadIdx.x & 63, l15 = lane & 15;
    typedef double __attribute__((address_space(1))) *gp_gptr;
    // (arguments of a real call arrive in vector registers: the image's address back into scalar ones)
    const unsigned long long d_bits = (unsigned long long)D;
    const unsigned long long d_uni = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(d_bits >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)d_bits);
    const char __attribute__((address_space(1))) *Dg = (const char __attribute__((address_space(1))) *)d_uni;
    auto gd = [&](unsigned uniform_off, unsigned lane_off) -> double __attribute__((address_space(1))) &
    { return *(gp_gptr)(dg_scalar_base(Dg + uniform_off) + dg_lane_offset(lane_off)); };
    auto tile_off = [&](int ti, int tj) -> unsigned
    { return (unsigned)((tj * 16) * nb + ti * 16) * 8u; };
    unsigned long long ph_ = dbg ? __builtin_amdgcn_s_memtime() : 0;
#define GP_PH(slot)                                                       \
    if (dbg && lane == 0 && blockIdx.x == 0)                              \
    {                                                                     \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
        dbg[slot] += now_ - ph_;                                          \
        ph_ = now_;                                                       \
    }
    double *dsel = rdiag + 16; // [0][r]: reciprocal pivot 15 - r (U group), [1][r]: 1 (L group)
    if (lane < 16)
        dsel[16 + lane] = 1.0;
    // ================= the factorisation wavefront: LU of diagonal tile k, then the inverses of its two factors =================
    for (int k = 0; k < NT; k++)
    {
        gp_barrier(); // A: the diagonal tile is in Td
        GP_PH(16)
        double x[16];
#pragma unroll
        for (int c = 0; c < 16; c++)
            x[c] = Td[l15][c]; // (lanes 16.. repeat rows 0..15; only lanes 0..15 write back)
        // Sixteen pivot steps as ONE basic block (no branches: rows at or above the pivot take a zero multiplier), so that the
        // scheduler can run the NEXT pivot's reciprocal -- a chain of seven dependent operations -- beside the bulk of the current
        // rank-1 update: the next pivot entry is updated and broadcast first.
        double myrp = 1.0;
        double pnext = gt_readlane(x[0], 0);
#pragma unroll
        for (int kk = 0; kk < 16; kk++)
        {
            double p = pnext;
            if ((p < 0 ? -p : p) < PANGULU_TOL)
                p = PANGULU_TOL;
            const double rp = gt_refined_rcp(p);
            myrp = (l15 == kk) ? rp : myrp;
            double u[16];
#pragma unroll
            for (int c = 0; c < 16; c++)
                if (c > kk)
                    u[c] = gt_readlane(x[c], kk);
            const bool below = l15 > kk;
            double l = x[kk] * rp;
            l = __builtin_fma(__builtin_fma(-l, p, x[kk]), rp, l); // one correction: the quotient to the last place
            l = below ? l : 0.0;
            x[kk] = below ? l : x[kk];
            if (kk + 1 < 16)
            {
                x[kk + 1] = __builtin_fma(-l, u[kk + 1], x[kk + 1]);
                pnext = gt_readlane(x[kk + 1], kk + 1);
            }
#pragma unroll
            for (int c = 0; c < 16; c++)
                if (c > kk + 1)
                    x[c] = __builtin_fma(-l, u[c], x[c]);
        }
        if (lane < 16)
        {
            rdiag[lane] = myrp;
            dsel[15 - lane] = myrp; // (the U group's row scaling in the index-reversed inversion below)
        }
        if (lane < 16)
        {
#pragma unroll
            for (int c = 0; c < 16; c++)
            {
                Mi[lane * 17 + c] = x[c];                        // plain image (the L factor is read from it)
                Mi[272 + (15 - lane) * 17 + (15 - c)] = x[c];    // index-reversed image (U becomes lower triangular)
            }
        }
        wave_lds_fence();
        GP_PH(17)
        // inverses: lanes 16..31 solve L y = e_c on the plain image, lanes 0..15 solve U x = e_c on the reversed one (a forward
        // substitution as well); the whole column goes to LDS (zeros outside the triangle)
        if (lane < 32)
        {
            const int grp = (lane >> 4) & 1, c = l15;
            const double *M = Mi + (grp ? 0 : 1) * 272;
            const double *dsl = dsel + grp * 16; // row scalings: 1 for L (unit diagonal), reciprocal pivots in reversed order for U
            const int cc = grp ? c : 15 - c;
            // Row by row with the NEXT row's entries of the factor in flight while the current one is summed (two row buffers; a
            // compiler barrier between the rows keeps the loads where they are: left alone, the compiler hoists all 120 entries to
            // the top, spills, and the inversion took 8 us per tile instead of the half microsecond its dependent chain needs).
            double z[16], bufA[16], bufB[16];
            z[0] = ((cc == 0) ? 1.0 : 0.0) * dsl[0];
            bufA[0] = M[1 * 17 + 0];
#pragma unroll
            for (int r = 1; r < 16; r++)
            {
                asm volatile("" ::: "memory");
                if (r + 1 < 16)
                {
#pragma unroll
                    for (int m = 0; m < 16; m++)
                        if (m < r + 1)
                        {
                            if (r & 1)
                                bufB[m] = M[(r + 1) * 17 + m];
                            else
                                bufA[m] = M[(r + 1) * 17 + m];
                        }
                }
                const double dr = dsl[r];
                double s0 = (r == cc) ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
                for (int m = 0; m < r; m++)
                {
                    const double mv = (r & 1) ? bufA[m] : bufB[m];
                    if (m & 1)
                        s1 = __builtin_fma(-mv, z[m], s1);
                    else
                        s0 = __builtin_fma(-mv, z[m], s0);
                }
                z[r] = (s0 + s1) * dr;
                asm volatile("" : "+v"(z[r])::"memory"); // (anchors this row's sums in front of the loads of the row after the next)
            }
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                if (grp)
                    IL[r][c] = z[r];
                else
                    IU[15 - r][c] = z[r];
            }
            // ... and into the image in memory, over the diagonal tile (U11^-1 on and above the diagonal, L11^-1 below): scalar
            // base + 32-bit lane offset (as 64-bit pointers these sixteen stores became sixteen induction variables in scratch)
            const unsigned col_lane = (unsigned)c * colB;
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                const int rowL = r, rowU = 15 - r;
                if (grp ? (rowL > c) : (rowU <= c))
                    gd(tile_off(k, k), col_lane + (unsigned)(grp ? rowL : rowU) * 8u) = z[r];
            }
        }
        if (lane < 16)
        {
            // the factorised tile goes behind the mirror for the deferred sparsify job
            const unsigned saved_off = (unsigned)(((size_t)nb * nb + MIRROR_MAP_BYTES / sizeof(double)) * 8u) + ((unsigned)k << 11);
#pragma unroll
            for (int c = 0; c < 16; c++)
                gd(saved_off + (unsigned)c * 128u, (unsigned)lane * 8u) = x[c]; // saved[(k << 8) + (c << 4) + lane]
        }
        GP_PH(18)
        gp_barrier(); // B: IL, IU ready
        GP_PH(19)
        gp_barrier(); // C
        GP_PH(20)
    }
#undef GP_PH
}

template <int NT>
__global__ __launch_bounds__(GP_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void getrf_pipe_f64_kernel(const GetrfTaskD *__restrict__ tasks, int nb_,
                                                                                                               unsigned long long *flop_counter, unsigned long long *dbg)
{
    constexpr int nb = NT * 16;
    constexpr int M0 = NT > 12 ? NT - 12 : 0;      // tiles with min(i, j) >= M0 are resident
    constexpr int RR = (NT - M0 + 1) / 2;          // resident tile rows of a trailing wavefront (2 x 3 grid)
    constexpr int RC = (NT - M0 + 2) / 3;          // resident tile columns
    constexpr int AR = (NT + 1) / 2, AC = (NT + 2) / 3; // all tile rows / columns of a trailing wavefront (band included)
    constexpr int ldp = nb + 2;
    (void)nb_;
    const GetrfTaskD T = tasks[blockIdx.x];
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double *Pm = reinterpret_cast<double *>(smem_raw); // Pm[c * ldp + r]: column c of the panel, row r (absolute)
    double *Sm = Pm + 16 * ldp;                        // Sm[k * ldp + c]: row k of the strip, column c (absolute)
    double(*Td)[17] = reinterpret_cast<double(*)[17]>(Sm + 16 * ldp); // diagonal tile: pre-LU, then L11 \ U11
    double *Mi = reinterpret_cast<double *>(Td) + 16 * 17;              // inversion images: [plain, index-reversed][16][17]
    double(*IL)[17] = reinterpret_cast<double(*)[17]>(Mi + 2 * 16 * 17); // L11^-1 [row][column]
    double(*IU)[17] = IL + 16;                                          // U11^-1 [row][column]
    double *rdiag = reinterpret_cast<double *>(IU + 16);                 // reciprocals of the (clamped) pivots of the tile
    u32 *sLcp = reinterpret_cast<u32 *>(rdiag + 48);
    u32 *sUrp = sLcp + nb + 1;
    unsigned *smap = sUrp + nb + 1; // smap[tj] bit ti: tile (ti, tj) holds pattern entries
    double *__restrict__ D = reinterpret_cast<double *>(T.dense);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    unsigned long long stamp_ = dbg ? __builtin_amdgcn_s_memtime() : 0;

    // ---- prologue: dense image + occupancy map (as the tiled kernel) -------------------------------------------------
    if (tid < 16)
        smap[tid] = 0;
    if (!T.preloaded)
    {
        for (int i = tid; i < nb * nb / 2; i += GP_THREADS)
            reinterpret_cast<double2 *>(D)[i] = make_double2(0.0, 0.0);
    }
    for (int i = tid; i <= nb; i += GP_THREADS)
    {
        sLcp[i] = T.lcp[i];
        sUrp[i] = T.urp[i];
    }
    __syncthreads();
    const u32 nnzL = sLcp[nb], nnzU = sUrp[nb];
    unsigned long long ops = 0;
    if (!T.preloaded)
    {
        for (u32 p = tid; p < nnzL; p += GP_THREADS)
        {
            const int c = owner_of(sLcp, nb, p);
            const u32 r = T.lri[p];
            D[(size_t)c * nb + r] = T.lval[p];
            atomicOr(&smap[c >> 4], 1u << (r >> 4));
        }
        for (u32 p = tid; p < nnzU; p += GP_THREADS)
        {
            const int r = owner_of(sUrp, nb, p);
            const u32 c = T.uci[p];
            D[(size_t)c * nb + r] = T.uval[p];
            atomicOr(&smap[c >> 4], 1u << (r >> 4));
        }
    }
    else if (tid < 16)
        smap[tid] = tid < NT ? (unsigned)mirror_map(D, nb)[tid] : 0u;
    for (int c = tid; c < nb; c += GP_THREADS)
    {
        // structural flop count of the sparse algorithm (what the reference counts, src/pangulu_kernel_interface.c:4-82)
        const u32 nl = sLcp[c + 1] - sLcp[c], nu = sUrp[c + 1] - sUrp[c];
        if (nu > 0)
            ops += (unsigned long long)nl * (1ull + 2ull * (nu - 1));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (a scattered image is read by other wavefronts below)
    __syncthreads();
    // the image is a mirror: the dense solves of this level skip structurally empty factor tiles by this map
    if (tid < 16)
        reinterpret_cast<unsigned short *>(D + (size_t)nb * nb)[tid] = (unsigned short)smap[tid];
    ops = wave_sum(ops);
    if (lane == 0 && ops)
        atomicAdd(flop_counter, ops);
    GETRF_STAMP(0)
    if (wave == 1 + GP_TWAVES)
        return; // (the eighth wavefront has no role: a finished wavefront does not count at the barriers below)

    // 32-bit byte offsets into the image (nb <= 256: below 512 KiB): one uniform base + one VGPR per access
    constexpr unsigned colB = (unsigned)nb * 8u, col4B = 4u * colB;
    (void)colB;
    const unsigned cf = ((unsigned)l4 * (unsigned)nb + (unsigned)l15) * 8u; // column form: register g of lane l is (r0 + l15, c0 + l4 + 4g)
    const unsigned rf = ((unsigned)l15 * (unsigned)nb + (unsigned)l4) * 8u; // row form:    register g of lane l is (r0 + l4 + 4g, c0 + l15)
    // a workgroup-uniform byte offset into the image on the scalar side + a 32-bit lane offset: the scalar-base form of global_load /
    // global_store (one address register per access instead of a 64-bit pair; see dg_scalar_base in pg_hip_dense.h)
    typedef double __attribute__((address_space(1))) *gp_gptr;
    const char __attribute__((address_space(1))) *Dg = (const char __attribute__((address_space(1))) *)reinterpret_cast<const char *>(D);
    auto gd = [&](unsigned uniform_off, unsigned lane_off) -> double __attribute__((address_space(1))) &
    { return *(gp_gptr)(dg_scalar_base(Dg + uniform_off) + dg_lane_offset(lane_off)); };
    auto tile_off = [&](int ti, int tj) -> unsigned
    { return (unsigned)((tj * 16) * nb + ti * 16) * 8u; };
    // LDS images by byte address (address space 3: 32-bit addresses, immediates for compile-time parts)
    typedef double __attribute__((address_space(3))) *gp_lptr;
    const unsigned pm_base = (unsigned)(unsigned long long)(void __attribute__((address_space(3))) *)Pm;
    constexpr unsigned SM_OFF = 16u * (unsigned)ldp * 8u; // Sm behind Pm
    const unsigned frag_lane = pm_base + ((unsigned)l4 * (unsigned)ldp + (unsigned)l15) * 8u; // + q * 4 ldp * 8 + tile * 128 (+ SM_OFF)
    const unsigned tstore_lane = pm_base + SM_OFF + ((unsigned)l15 * (unsigned)ldp + (unsigned)l4) * 8u; // strip image, transposed store: + tile * 128 + g * 32
    // (an address formed WHERE it is used -- lane constant + tile offset, one vector add -- instead of one register per (tile, quarter)
    //  of the wavefront kept for the whole kernel: the first build spilled 46 of them and reloaded them from scratch in every step)
    auto fresh = [](unsigned v) -> unsigned
    {
        asm volatile("" : "+v"(v));
        return v;
    };

    // phase stamps (PANGULU_HIP_DEBUG_GETRF / tools/microbench/bench_getrf.hip): block 0; slots 16.. by wavefront 0 (LU, inverses, waits),
    // 24.. by trailing wavefront 1 (trailing, finish, next diagonal tile, waits at the three barriers)
    unsigned long long ph_ = dbg ? __builtin_amdgcn_s_memtime() : 0;
#define GP_PH(slot)                                                       \
    if (dbg && lane == 0 && blockIdx.x == 0)                              \
    {                                                                     \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
        dbg[slot] += now_ - ph_;                                          \
        ph_ = now_;                                                       \
    }
    if (wave == 0)
    {
        // the factorisation wavefront runs in a function of its own: inlined, its sixteen pivot steps and the two inversions shared one
        // register allocation with the trailing wavefronts' 192 resident registers and spilled into scratch inside the chain
        gp_factor_wavefront<NT>(D, Td, Mi, IL, IU, rdiag, dbg);
        return;
    }

    // ================= trailing wavefronts =================================================================================
    const int tw = wave - 1, tr = tw & 1, tc = tw >> 1;
    auto row_of = [&](int ri) -> int { return NT - 1 - tr - 2 * ri; };
    auto col_of = [&](int ci) -> int { return NT - 1 - tc - 3 * ci; };
    // occupancy of this wavefront's tile columns (bit i of colmap[ci]: tile (i, col_of(ci)) holds pattern entries)
    unsigned colmap[AC];
#pragma unroll
    for (int ci = 0; ci < AC; ci++)
        colmap[ci] = col_of(ci) >= 0 ? (unsigned)__builtin_amdgcn_readfirstlane((int)smap[col_of(ci) < 0 ? 0 : col_of(ci)]) : 0u;

    v4f64 R[RR][RC]; // resident tiles, column form
#pragma unroll
    for (int ri = 0; ri < RR; ri++)
#pragma unroll
        for (int ci = 0; ci < RC; ci++)
            R[ri][ci] = (v4f64){0.0, 0.0, 0.0, 0.0};

    // a tile of the image in memory, column form
    auto load_tile = [&](int ti, int tj) -> v4f64
    {
        v4f64 t;
        const unsigned o = tile_off(ti, tj);
#pragma unroll
        for (int g = 0; g < 4; g++)
            t[g] = gd(o + g * col4B, cf);
        return t;
    };
    auto store_tile = [&](int ti, int tj, const v4f64 &t)
    {
        const unsigned o = tile_off(ti, tj);
#pragma unroll
        for (int g = 0; g < 4; g++)
            gd(o + g * col4B, cf) = t[g];
    };
    // ... row form: as the first MFMA source of a product (a finished U tile read back for a band tile's left-looking updates)
    auto load_tile_rowform = [&](int ti, int tj) -> v4f64
    {
        v4f64 t;
        const unsigned o = tile_off(ti, tj);
#pragma unroll
        for (int g = 0; g < 4; g++)
            t[g] = gd(o + 32u * g, rf);
        return t;
    };
    // MFMA operand fragments of k-quarter q from the images: of the panel for tile row ti, of the strip for tile column tj
    auto panel_frag = [&](int q, int ti) -> double
    { return *(gp_lptr)(unsigned long long)(fresh(frag_lane) + (unsigned)(ti < 0 ? 0 : ti) * 128u + (unsigned)q * (4u * (unsigned)ldp * 8u)); };
    auto strip_frag = [&](int q, int tj) -> double
    { return *(gp_lptr)(unsigned long long)(fresh(frag_lane) + (unsigned)(tj < 0 ? 0 : tj) * 128u + (SM_OFF + (unsigned)q * (4u * (unsigned)ldp * 8u))); };
    // band tile (min(ti, tj) < M0): everything the steps 0 .. upto-1 owe it, from the finished factor tiles in memory
    auto band_catch_up = [&](v4f64 t, int ti, int tj, int upto) -> v4f64
    {
        for (int m = 0; m < upto; m++)
        {
            const unsigned cm = (unsigned)__builtin_amdgcn_readfirstlane((int)smap[m]);
            if (!((cm >> ti) & 1u) || !((unsigned)__builtin_amdgcn_readfirstlane((int)smap[tj]) >> m & 1u))
                continue;
            const v4f64 a = load_tile(ti, m), b = load_tile_rowform(m, tj);
#pragma unroll
            for (int q = 0; q < 4; q++)
                t = __builtin_amdgcn_mfma_f64_16x16x4f64(b[q], a[q], t, 0, 0, DG_NEG_A);
        }
        return t;
    };
    // finish a panel tile: X = T U11^-1 (the tile is the second MFMA source as it stands), into the panel image and into memory
    auto finish_panel = [&](v4f64 t, int ti, int k) -> v4f64
    {
        v4f64 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; q++)
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(IU[4 * q + l4][l15], t[q], x, 0, 0, 0);
        const unsigned pa = fresh(frag_lane) + (unsigned)ti * 128u;
#pragma unroll
        for (int g = 0; g < 4; g++)
            *(gp_lptr)(unsigned long long)(pa + (unsigned)g * (4u * (unsigned)ldp * 8u)) = x[g]; // Pm[(l4 + 4 g) * ldp + ti * 16 + l15]
        store_tile(ti, k, x);
        return x;
    };
    // finish a strip tile: Y = L11^-1 T; the tile has to be the FIRST source, i.e. transposed in the lanes: through the strip image
    auto finish_strip = [&](v4f64 t, int k, int tj) -> v4f64
    {
        const unsigned sa = fresh(tstore_lane) + (unsigned)tj * 128u; // Sm[l15 * ldp + tj * 16 + l4 + 4 g]
#pragma unroll
        for (int g = 0; g < 4; g++)
            *(gp_lptr)(unsigned long long)(sa + 32u * (unsigned)g) = t[g];
        wave_lds_fence();
        double f[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
            f[q] = strip_frag(q, tj);
        v4f64 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; q++)
            y = __builtin_amdgcn_mfma_f64_16x16x4f64(f[q], IL[l15][4 * q + l4], y, 0, 0, 0);
        wave_lds_fence();
#pragma unroll
        for (int g = 0; g < 4; g++)
            *(gp_lptr)(unsigned long long)(sa + 32u * (unsigned)g) = y[g];
        store_tile(k, tj, y);
        return y;
    };

    // diagonal tile 0 to its owner's hands ... and into Td
    if (tr == ((NT - 1) & 1) && tc == (NT - 1) % 3)
    {
        const v4f64 t = load_tile(0, 0);
#pragma unroll
        for (int g = 0; g < 4; g++)
            Td[l15][l4 + 4 * g] = t[g];
    }
    gp_barrier(); // A (step 0)
    // the resident tiles, while wavefront 0 factorises the first diagonal tile
#pragma unroll
    for (int ri = 0; ri < RR; ri++)
#pragma unroll
        for (int ci = 0; ci < RC; ci++)
        {
            const int i = row_of(ri), j = col_of(ci);
            if (i >= M0 && j >= M0 && ((colmap[ci] >> i) & 1u))
                R[ri][ci] = load_tile(i, j);
        }

    for (int k = 0; k < NT; k++)
    {
        const unsigned colk = (unsigned)__builtin_amdgcn_readfirstlane((int)smap[k]); // bit i: tile (i, k) holds pattern entries
        // ---- between A and B: panel k - 1 on the resident tiles (the diagonal tile k has had it already) -----------------
        if (k > 0)
        {
            const int kp = k - 1;
            const unsigned colp = (unsigned)__builtin_amdgcn_readfirstlane((int)smap[kp]);
            // this wavefront's resident rows / columns beyond panel kp form a rectangle anchored at index 0
            unsigned rowlive = 0, collive = 0;
#pragma unroll
            for (int ri = 0; ri < RR; ri++)
                if (row_of(ri) > kp && row_of(ri) >= M0 && ((colp >> row_of(ri)) & 1u))
                    rowlive |= 1u << ri;
#pragma unroll
            for (int ci = 0; ci < RC; ci++)
                if (col_of(ci) > kp && col_of(ci) >= M0 && ((colmap[ci] >> kp) & 1u))
                    collive |= 1u << ci;
            if (rowlive && collive)
            {
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    // (fragments: the strip's once per quarter, the panel's one tile row ahead of the matrix cores -- few registers
                    //  beside the 192 the resident tiles take)
                    double b[RC];
#pragma unroll
                    for (int ci = 0; ci < RC; ci++)
                        b[ci] = strip_frag(q, col_of(ci));
                    double a_cur = panel_frag(q, row_of(0));
#pragma unroll
                    for (int ri = 0; ri < RR; ri++)
                    {
                        double a_nxt = a_cur;
                        if (ri + 1 < RR)
                            a_nxt = panel_frag(q, row_of(ri + 1));
                        if ((rowlive >> ri) & 1u)
                        {
#pragma unroll
                            for (int ci = 0; ci < RC; ci++)
                            {
                                // (the next diagonal tile took this panel before barrier A)
                                if (((collive >> ci) & 1u) && !(row_of(ri) == k && col_of(ci) == k))
                                    R[ri][ci] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[ci], a_cur, R[ri][ci], 0, 0, DG_NEG_A);
                            }
                        }
                        a_cur = a_nxt;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        if (wave == 1)
        {
            GP_PH(24)
        }
        gp_barrier(); // B: L11^-1 and U11^-1 of step k are in LDS; nobody reads the images of step k - 1 any more
        if (wave == 1)
        {
            GP_PH(25)
        }
        // ---- between B and C: this wavefront's tiles of tile column k (panel) and tile row k (strip) are finished ---------
        {
            const int ck = NT - 1 - tc - k; // column index ci with col_of(ci) == k, times 3
            if (ck >= 0 && ck % 3 == 0)
            {
                const int cik = ck / 3;
#pragma unroll
                for (int ri = 0; ri < AR; ri++)
                {
                    const int i = row_of(ri);
                    if (i <= k || !((colk >> i) & 1u))
                        continue;
                    if (k >= M0)
                    {
                        // resident (ri < RR, cik < RC): static register indices by enumeration
                        if (ri < RR)
                        {
#pragma unroll
                            for (int ci = 0; ci < RC; ci++)
                                if (ci == cik)
                                    R[ri][ci] = finish_panel(R[ri][ci], i, k);
                        }
                    }
                    else
                        finish_panel(band_catch_up(load_tile(i, k), i, k, k), i, k);
                }
            }
            const int rk = NT - 1 - tr - k; // row index ri with row_of(ri) == k, times 2
            if (rk >= 0 && rk % 2 == 0)
            {
                const int rik = rk / 2;
#pragma unroll
                for (int ci = 0; ci < AC; ci++)
                {
                    const int j = col_of(ci);
                    if (j <= k || j < 0 || !((colmap[ci] >> k) & 1u))
                        continue;
                    if (k >= M0)
                    {
                        if (ci < RC)
                        {
#pragma unroll
                            for (int ri = 0; ri < RR; ri++)
                                if (ri == rik)
                                    R[ri][ci] = finish_strip(R[ri][ci], k, j);
                        }
                    }
                    else
                        finish_strip(band_catch_up(load_tile(k, j), k, j, k), k, j);
                }
            }
        }
        if (k + 1 < M0 + 1)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (band steps: finished tiles are read back from memory by other wavefronts)
        if (wave == 1)
        {
            GP_PH(26)
        }
        gp_barrier(); // C: the images of step k are complete
        if (wave == 1)
        {
            GP_PH(27)
        }
        // ---- between C and A: the next diagonal tile takes panel k and goes to the factorisation wavefront ------------------
        if (k + 1 < NT)
        {
            const int kn = k + 1;
            const int rn = NT - 1 - tr - kn, cn = NT - 1 - tc - kn;
            if (rn >= 0 && rn % 2 == 0 && cn >= 0 && cn % 3 == 0)
            {
                const int rin = rn / 2, cin = cn / 3;
                const bool reaches = ((colk >> kn) & 1u) && (((unsigned)__builtin_amdgcn_readfirstlane((int)smap[kn]) >> k) & 1u);
                v4f64 t;
                if (kn >= M0)
                {
                    t = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int ri = 0; ri < RR; ri++)
#pragma unroll
                        for (int ci = 0; ci < RC; ci++)
                            if (ri == rin && ci == cin)
                                t = R[ri][ci];
                    if (reaches)
                    {
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            t = __builtin_amdgcn_mfma_f64_16x16x4f64(strip_frag(q, kn), panel_frag(q, kn), t, 0, 0, DG_NEG_A);
                    }
                }
                else
                    t = band_catch_up(load_tile(kn, kn), kn, kn, kn); // (a band tile: all of steps 0 .. k at once, from memory)
#pragma unroll
                for (int g = 0; g < 4; g++)
                    Td[l15][l4 + 4 * g] = t[g];
            }
            if (wave == 1)
            {
                GP_PH(28)
            }
            gp_barrier(); // A (step k + 1)
            if (wave == 1)
            {
                GP_PH(29)
            }
        }
    }
    GETRF_STAMP(5)
#undef GP_PH
}
