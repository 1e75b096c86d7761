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
//   * the band of the first tile rows / columns that does not fit (tiles with min(i, j) < 4 at nb = 256) takes the panels
//     right-looking as well, but THROUGH MEMORY: in the trailing phase of step k its owner loads a band tile (three in flight),
//     applies panel k - 1 from the same LDS images and stores it back -- except the tiles of step k's own panel and strip, which go
//     to STAGING SLOTS in LDS (2 x 16 slots of [16][17] doubles) and are finished from there.  (The first version was left-looking:
//     a band tile received all its updates at the step that finishes it, from factor tiles read back from memory -- up to three
//     dependent trips to L2 per tile inside the one phase every wavefront waits for: 75 of 198 us.)
//     Also tried: the idle eighth wavefront doing ALL band updates with eight tiles in flight (profiles/r05y_*): 185 us against 153 --
//     it shares its SIMD with a trailing wavefront whose f64 MFMAs hold up every vector instruction it issues, and 185 band-tile
//     updates of 0.4 us each in one wavefront is longer than the 31 the six owners do each.
//   * panel and strip tiles are finished by their OWNERS on the matrix cores with the inverses of the diagonal tile's factors
//     (X = T U11^-1, Y = L11^-1 T; the accumulator layout of a tile is the operand layout of the second MFMA source, so a resident
//     tile is an operand as it stands); those inverses are what the dense TSTRF/GESSM of the level want in the image anyway
//     (pg_hip_trsm_dense.h), so nothing is computed twice, and the 16-step substitutions are gone.
//   * ONE wavefront factorises the diagonal tile (branch-free, the next pivot's reciprocal beside the current rank-1 update) and
//     inverts its two factors BLOCKED BY 4 ON THE MATRIX CORES (W' = W - (W B) W per level, see there) while the others apply the
//     previous panel to the trailing block; three workgroup barriers per step:
//         A  diagonal tile k in LDS      | wavefront 0: LU + inverses        ||  trailing wavefronts: panel k-1 on their tiles
//         B  inverses in LDS, images free| owners finish panel / strip tiles of step k into the images and the block's image in memory
//         C  images of step k complete   | the owner of diagonal tile k+1 applies panel k to it and hands it over
//   * LDS: 47 KB at nb = 128, 148 KB at nb = 256 (one image pair + the staging slots).  The 256 registers
//     per wavefront allow one workgroup per CU whatever the LDS says.
// Order of operations per entry: updates in ascending pivot order panel by panel, summed by the matrix cores inside a panel (as the
// tiled kernel); the panel solves multiply by explicit 16 x 16 inverses instead of substituting (as the dense TSTRF/GESSM do).
// Parity: within 1e-12 of the oracle (tests/test_gpu_parity*.py run every case on this kernel by default; PANGULU_HIP_GETRF_PIPE=0
// selects the tiled kernel, which also still serves images that are not mirrors or whose sparse record is gathered in the kernel).
#pragma once

#define GP_THREADS 512
#define GP_TWAVES 6
#ifndef GP_IDLE_WAVE
#define GP_IDLE_WAVE 4 // of the eight wavefronts: 0 factorises, six trail, this one leaves
#endif

#define GP_STAGE_SLOT (16 * 17 * 8) // bytes of one staged band tile ([16][17] doubles)
__host__ __device__ inline size_t gp_lds_bytes(int nb)
{
    const size_t images = sizeof(double) * (2 * 16 * (size_t)(nb + 2) + 16 * 17 /* Td */ + 3 * 16 * 17 /* factorised tile, block-diagonal inverses of L and U */ + 2 * 16 * 17 /* IL, IU */ + 48 /* rdiag, row scalings of the inversion */) +
                          sizeof(unsigned) * (2 * (size_t)(nb + 1) + 16);
    // (nb = 256: the panel and strip tiles of a band step wait in LDS between their last update and their finish)
    return ((images + 15) & ~(size_t)15) + (nb > 192 ? (size_t)2 * (nb / 16) * GP_STAGE_SLOT : 0);
}

#ifndef GP_BAND_CHUNK
#define GP_BAND_CHUNK 3 // band tiles in flight per trailing wavefront
#endif
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
    const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
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
    for (int i = lane; i < 2 * 272; i += 64)
        Mi[272 + i] = 0.0; // (the two block-diagonal images: only their diagonal blocks are ever written)
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
        }
        if (lane < 16)
        {
#pragma unroll
            for (int c = 0; c < 16; c++)
                Mi[lane * 17 + c] = x[c];
        }
        wave_lds_fence();
        GP_PH(17)
        // Inverses of the two factors, blocked by 4 on the matrix cores.  With W0 = the inverses of the four 4 x 4 diagonal blocks
        // (block diagonal) and B the factor's blocks OFF that diagonal, one level doubles the block size:
        //       W' = W - (W B') W ,  B' = the off-diagonal blocks that pair up neighbouring diagonal blocks of this level
        // ([[A, 0], [B, C]]^-1 = [[A^-1, 0], [-C^-1 B A^-1, C^-1]] for L; the mirrored statement for U): two levels, each two
        // dependent products of which only two k-quarters are non-zero (a quarter of the MFMA's k IS a 4-block).  The products
        // chain in registers where the previous result is the left factor (the accumulator layout is the second source's);
        // W goes through LDS once per level to become a right factor.  L and U run interleaved.  0.9 us per tile where the
        // sixteen-step substitutions of the previous build took 3.0.
        {
            const int grp = (lane >> 4) & 1, b4 = (l15 >> 2) * 4, cb = l15 & 3; // lanes 0..15: U (index-reversed), 16..31: L
            double(*W0L)[17] = reinterpret_cast<double(*)[17]>(Mi + 272);
            double(*W0U)[17] = reinterpret_cast<double(*)[17]>(Mi + 544);
            // first-source form of the whole factorised tile: M(4 q + l4, l15)
            double mq[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
                mq[q] = Mi[(4 * q + l4) * 17 + l15];
            if (lane < 32)
            {
                // column cb of the inverse of diagonal block b4 / 4: forward substitution (U through reversed indices)
                const int base = grp ? (b4 * 17 + b4) : ((b4 + 3) * 17 + b4 + 3), sg = grp ? 1 : -1;
                const double m10 = Mi[base + sg * 17], m20 = Mi[base + sg * 34], m21 = Mi[base + sg * 35];
                const double m30 = Mi[base + sg * 51], m31 = Mi[base + sg * 52], m32 = Mi[base + sg * 53];
                double d0 = 1.0, d1 = 1.0, d2 = 1.0, d3 = 1.0;
                if (!grp)
                {
                    d0 = rdiag[b4 + 3];
                    d1 = rdiag[b4 + 2];
                    d2 = rdiag[b4 + 1];
                    d3 = rdiag[b4];
                }
                const int cc = grp ? cb : 3 - cb;
                const double z0 = (cc == 0 ? 1.0 : 0.0) * d0;
                const double z1 = __builtin_fma(-m10, z0, cc == 1 ? 1.0 : 0.0) * d1;
                const double z2 = __builtin_fma(-m21, z1, __builtin_fma(-m20, z0, cc == 2 ? 1.0 : 0.0)) * d2;
                const double z3 = __builtin_fma(-m32, z2, __builtin_fma(-m31, z1, __builtin_fma(-m30, z0, cc == 3 ? 1.0 : 0.0))) * d3;
                double *W = grp ? &W0L[b4][b4 + cb] : &W0U[b4 + 3][b4 + cb];
                W[0] = z0;
                W[sg * 17] = z1;
                W[sg * 34] = z2;
                W[sg * 51] = z3;
            }
            wave_lds_fence();
            v4f64 wl, wu; // W in column form (= second source, k-quarter by register)
            double fl[4], fu[4]; // W in first-source form
#pragma unroll
            for (int q = 0; q < 4; q++)
            {
                wl[q] = W0L[l15][4 * q + l4];
                wu[q] = W0U[l15][4 * q + l4];
                fl[q] = W0L[4 * q + l4][l15];
                fu[q] = W0U[4 * q + l4][l15];
            }
            const int cblk = l15 >> 2;
            const v4f64 zero4 = {0.0, 0.0, 0.0, 0.0};
            // level 1: blocks (1,0), (3,2) of L and (0,1), (2,3) of U
            v4f64 tl = zero4, tu = zero4;
            tl = __builtin_amdgcn_mfma_f64_16x16x4f64(cblk == 0 ? mq[1] : 0.0, wl[1], tl, 0, 0, 0);
            tu = __builtin_amdgcn_mfma_f64_16x16x4f64(cblk == 1 ? mq[0] : 0.0, wu[0], tu, 0, 0, 0);
            tl = __builtin_amdgcn_mfma_f64_16x16x4f64(cblk == 2 ? mq[3] : 0.0, wl[3], tl, 0, 0, 0);
            tu = __builtin_amdgcn_mfma_f64_16x16x4f64(cblk == 3 ? mq[2] : 0.0, wu[2], tu, 0, 0, 0);
            wl = __builtin_amdgcn_mfma_f64_16x16x4f64(fl[0], tl[0], wl, 0, 0, DG_NEG_A);
            wu = __builtin_amdgcn_mfma_f64_16x16x4f64(fu[1], tu[1], wu, 0, 0, DG_NEG_A);
            wl = __builtin_amdgcn_mfma_f64_16x16x4f64(fl[2], tl[2], wl, 0, 0, DG_NEG_A);
            wu = __builtin_amdgcn_mfma_f64_16x16x4f64(fu[3], tu[3], wu, 0, 0, DG_NEG_A);
            // level 2: block (1,0) of 8 x 8 blocks of L, (0,1) of U; W of level 1 through IL / IU to become a right factor
#pragma unroll
            for (int g = 0; g < 4; g++)
            {
                IL[l15][l4 + 4 * g] = wl[g];
                IU[l15][l4 + 4 * g] = wu[g];
            }
            tl = zero4;
            tu = zero4;
            tl = __builtin_amdgcn_mfma_f64_16x16x4f64(l15 < 8 ? mq[2] : 0.0, wl[2], tl, 0, 0, 0);
            tu = __builtin_amdgcn_mfma_f64_16x16x4f64(l15 >= 8 ? mq[0] : 0.0, wu[0], tu, 0, 0, 0);
            tl = __builtin_amdgcn_mfma_f64_16x16x4f64(l15 < 8 ? mq[3] : 0.0, wl[3], tl, 0, 0, 0);
            tu = __builtin_amdgcn_mfma_f64_16x16x4f64(l15 >= 8 ? mq[1] : 0.0, wu[1], tu, 0, 0, 0);
            wave_lds_fence();
            const double gl0 = IL[l4][l15], gl1 = IL[4 + l4][l15], gu2 = IU[8 + l4][l15], gu3 = IU[12 + l4][l15];
            wl = __builtin_amdgcn_mfma_f64_16x16x4f64(gl0, tl[0], wl, 0, 0, DG_NEG_A);
            wu = __builtin_amdgcn_mfma_f64_16x16x4f64(gu2, tu[2], wu, 0, 0, DG_NEG_A);
            wl = __builtin_amdgcn_mfma_f64_16x16x4f64(gl1, tl[1], wl, 0, 0, DG_NEG_A);
            wu = __builtin_amdgcn_mfma_f64_16x16x4f64(gu3, tu[3], wu, 0, 0, DG_NEG_A);
            // the inverses: into LDS for the finishing wavefronts, and into the image in memory over the diagonal tile (U11^-1 on
            // and above the diagonal, L11^-1 below)
            const unsigned cfl = ((unsigned)l4 * (unsigned)nb + (unsigned)l15) * 8u;
#pragma unroll
            for (int g = 0; g < 4; g++)
            {
                IL[l15][l4 + 4 * g] = wl[g];
                IU[l15][l4 + 4 * g] = wu[g];
                gd(tile_off(k, k) + (unsigned)g * 4u * colB, cfl) = (l15 > l4 + 4 * g) ? wl[g] : wu[g];
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
    double *Mi = reinterpret_cast<double *>(Td) + 16 * 17;              // inversion images [3][16][17]: L \\ U, blockdiag(L)^-1, blockdiag(U)^-1
    double(*IL)[17] = reinterpret_cast<double(*)[17]>(Mi + 3 * 16 * 17); // L11^-1 [row][column]
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
    if (wave == GP_IDLE_WAVE)
        return; // (no role: a finished wavefront does not count at the barriers below.  It is the one that would share the factorisation
                //  wavefront's SIMD -- wavefront w runs on SIMD w % 4 -- and whose f64 MFMAs would hold up every instruction of that chain)

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
    // staging slots of a band step (nb = 256): slot i = panel tile (i, k) as [column][row], slot 16 + j = strip tile (k, j) as [row][column]
    const unsigned stage_base = pm_base + (unsigned)((gp_lds_bytes(nb) - (nb > 192 ? (size_t)2 * NT * GP_STAGE_SLOT : 0)));
    const unsigned stage_lane_cr = stage_base + ((unsigned)l4 * 17u + (unsigned)l15) * 8u; // + g * 4 * 17 * 8: element (l15, l4 + 4 g) of [c][r] / (l4 + 4 g, l15) of [r][c]
    const unsigned stage_lane_rc = stage_base + ((unsigned)l15 * 17u + (unsigned)l4) * 8u; // + g * 32:         element (l15, l4 + 4 g) of [r][c]
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
    const int tw = wave < GP_IDLE_WAVE ? wave - 1 : wave - 2, tr = tw & 1, tc = tw >> 1;
    const unsigned panel_base = frag_lane - (unsigned)tr * 128u;          // + tile row offsets of this wavefront: (NT - 1 - 2 ri) * 128
    const unsigned strip_base = frag_lane + SM_OFF - (unsigned)tc * 128u; // + (NT - 1 - 3 ci) * 128
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
    // (ONE lane-offset register per tile, four scalar bases: every vector instruction of a trailing wavefront is issued behind the
    //  f64 MFMAs of the wavefront it shares its SIMD with, so address arithmetic in vector registers is paid in matrix-pipe time)
    auto load_tile = [&](int ti, int tj) -> v4f64
    {
        v4f64 t;
        const unsigned o = tile_off(ti, tj);
        const unsigned lo = dg_lane_offset(cf);
#pragma unroll
        for (int g = 0; g < 4; g++)
            t[g] = *(gp_gptr)(dg_scalar_base(Dg + (o + g * col4B)) + lo);
        return t;
    };
    auto store_tile = [&](int ti, int tj, const v4f64 &t)
    {
        const unsigned o = tile_off(ti, tj);
        const unsigned lo = dg_lane_offset(cf);
#pragma unroll
        for (int g = 0; g < 4; g++)
            *(gp_gptr)(dg_scalar_base(Dg + (o + g * col4B)) + lo) = t[g];
    };
    // MFMA operand fragments of k-quarter q from the images: of the panel for tile row ti, of the strip for tile column tj
    auto panel_frag = [&](int q, int ti) -> double
    { return *(gp_lptr)(unsigned long long)(fresh(frag_lane) + (unsigned)(ti < 0 ? 0 : ti) * 128u + (unsigned)q * (4u * (unsigned)ldp * 8u)); };
    auto strip_frag = [&](int q, int tj) -> double
    { return *(gp_lptr)(unsigned long long)(fresh(frag_lane) + (unsigned)(tj < 0 ? 0 : tj) * 128u + (SM_OFF + (unsigned)q * (4u * (unsigned)ldp * 8u))); };
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

    // (a wait the COMPILER sees: with the resident loads still pending in its books it would drain the load counter in front of every
    //  matrix instruction of the loop below -- and with it the band chunks in flight there)
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    for (int k = 0; k < NT; k++)
    {
        const unsigned colk = (unsigned)__builtin_amdgcn_readfirstlane((int)smap[k]); // bit i: tile (i, k) holds pattern entries
        // ---- between A and B: panel k - 1 on this wavefront's tiles (the diagonal tile k has had it already) ------------------
        // Resident tiles: four MFMAs each on operands from the images, nothing travels.
        // Band tiles (min(i, j) < M0) take the panels right-looking as well, but through memory, GP_BAND_CHUNK tiles at a time:
        // lane u proposes tile (row_of(u / AC), col_of(u % AC)) and the live ones are walked by the ballot.  (Issuing a chunk's
        // loads in front of a k-quarter of the resident update and consuming them behind it was tried: with 250 registers live the
        // compiler reuses the chunk's registers inside the quarter and drains the load counter there -- no gain, 156 against 152 us.)
        // The tiles of THIS step's panel and strip (min(i, j) == k) go to the staging slots in LDS instead of back to memory:
        // their owner finishes them from there behind barrier B (in step 0 they pass through here untouched).
        {
            const int kp = k - 1;
            const unsigned colp = kp >= 0 ? (unsigned)__builtin_amdgcn_readfirstlane((int)smap[kp < 0 ? 0 : kp]) : 0u;
            unsigned long long live = 0, updm = 0;
            if (M0 > 0 && k < M0)
            {
                const int u_ri = lane / AC, u_ci = lane - u_ri * AC;
                const int u_i = row_of(u_ri), u_j = col_of(u_ci);
                const int u_m = u_i < u_j ? u_i : u_j;
                const unsigned u_col = smap[u_j < 0 ? 0 : u_j];
                const bool cand = lane < AR * AC && u_m >= k && u_m < M0 && !(u_i == k && u_j == k);
                const bool upd = cand && kp >= 0 && ((colp >> (u_i & 31)) & 1u) && ((u_col >> (kp & 31)) & 1u);
                const bool stg = cand && u_m == k && ((u_col >> (u_i & 31)) & 1u);
                live = __builtin_amdgcn_ballot_w64(upd || stg);
                updm = __builtin_amdgcn_ballot_w64(upd);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (this wavefront's stores of the previous step to the same tiles)
            }
            // this wavefront's resident rows / columns beyond panel kp form a rectangle anchored at index 0
            unsigned rowlive = 0, collive = 0;
            if (k > 0)
            {
#pragma unroll
                for (int ri = 0; ri < RR; ri++)
                    if (row_of(ri) > kp && row_of(ri) >= M0 && ((colp >> row_of(ri)) & 1u))
                        rowlive |= 1u << ri;
#pragma unroll
                for (int ci = 0; ci < RC; ci++)
                    if (col_of(ci) > kp && col_of(ci) >= M0 && ((colmap[ci] >> kp) & 1u))
                        collive |= 1u << ci;
            }
            const bool resident = rowlive && collive;
#define GP_CHUNK_LOAD()                                           \
    int ti[GP_BAND_CHUNK], tj[GP_BAND_CHUNK];                     \
    bool tu[GP_BAND_CHUNK];                                       \
    v4f64 t[GP_BAND_CHUNK];                                       \
    _Pragma("unroll") for (int s = 0; s < GP_BAND_CHUNK; s++)     \
    {                                                             \
        ti[s] = -1;                                               \
        tj[s] = 0;                                                \
        tu[s] = false;                                            \
        if (live)                                                 \
        {                                                         \
            const int u = __builtin_ctzll(live);                  \
            live &= live - 1;                                     \
            ti[s] = row_of(u / AC);                               \
            tj[s] = col_of(u % AC);                               \
            tu[s] = (updm >> u) & 1ull;                           \
            t[s] = load_tile(ti[s], tj[s]);                       \
        }                                                         \
    }
#define GP_CHUNK_PROCESS()                                                                                                                   \
    _Pragma("unroll") for (int s = 0; s < GP_BAND_CHUNK; s++) if (ti[s] >= 0)                                                                \
    {                                                                                                                                        \
        if (tu[s])                                                                                                                           \
        {                                                                                                                                    \
            /* (one address per operand tile, the k-quarters by immediate offsets) */                                                        \
            const unsigned pa_ = fresh(frag_lane) + (unsigned)ti[s] * 128u, sb_ = fresh(frag_lane) + (unsigned)tj[s] * 128u;                 \
            double fa_[4], fb_[4];                                                                                                           \
            _Pragma("unroll") for (int qq = 0; qq < 4; qq++)                                                                                 \
            {                                                                                                                                \
                fa_[qq] = *(gp_lptr)(unsigned long long)(pa_ + (unsigned)qq * (4u * (unsigned)ldp * 8u));                                    \
                fb_[qq] = *(gp_lptr)(unsigned long long)(sb_ + (SM_OFF + (unsigned)qq * (4u * (unsigned)ldp * 8u)));                         \
            }                                                                                                                                \
            _Pragma("unroll") for (int qq = 0; qq < 4; qq++)                                                                                 \
                t[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb_[qq], fa_[qq], t[s], 0, 0, DG_NEG_A);                                         \
        }                                                                                                                                    \
        if (tj[s] == k)                                                                                                                      \
        {                                                                                                                                    \
            /* panel tile: [column][row], as the finish reads it (second MFMA source = the accumulator layout) */                            \
            const unsigned sa = fresh(stage_lane_cr) + (unsigned)ti[s] * GP_STAGE_SLOT;                                                      \
            _Pragma("unroll") for (int g = 0; g < 4; g++) *(gp_lptr)(unsigned long long)(sa + (unsigned)g * (4u * 17u * 8u)) = t[s][g];       \
        }                                                                                                                                    \
        else if (ti[s] == k)                                                                                                                 \
        {                                                                                                                                    \
            /* strip tile: [row][column]; the finish reads it transposed in the lanes (first MFMA source) */                                 \
            const unsigned sa = fresh(stage_lane_rc) + (unsigned)(16 + tj[s]) * GP_STAGE_SLOT;                                               \
            _Pragma("unroll") for (int g = 0; g < 4; g++) *(gp_lptr)(unsigned long long)(sa + (unsigned)g * 32u) = t[s][g];                   \
        }                                                                                                                                    \
        else                                                                                                                                 \
            store_tile(ti[s], tj[s], t[s]);                                                                                                  \
    }
            if (M0 > 0)
            {
                while (live)
                {
                    GP_CHUNK_LOAD()
                    GP_CHUNK_PROCESS()
                }
            }
            if (resident)
            {
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    // (fragments: the strip's once per quarter, the panel's one tile row ahead of the matrix cores -- few registers
                    //  beside the 192 the resident tiles take)
                    // (addresses: one register per image for the whole kernel + compile-time offsets -- the tile indices of a wavefront
                    //  are constants minus its grid coordinates -- so these reads cost no vector instruction)
                    double b[RC];
#pragma unroll
                    for (int ci = 0; ci < RC; ci++)
                        b[ci] = *(gp_lptr)(unsigned long long)(strip_base + (unsigned)((NT - 1 - 3 * ci) * 128 + q * (4 * ldp * 8)));
                    double a_cur = *(gp_lptr)(unsigned long long)(panel_base + (unsigned)((NT - 1) * 128 + q * (4 * ldp * 8)));
#pragma unroll
                    for (int ri = 0; ri < RR; ri++)
                    {
                        double a_nxt = a_cur;
                        if (ri + 1 < RR)
                            a_nxt = *(gp_lptr)(unsigned long long)(panel_base + (unsigned)((NT - 1 - 2 * (ri + 1)) * 128 + q * (4 * ldp * 8)));
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
#undef GP_CHUNK_LOAD
#undef GP_CHUNK_PROCESS
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
        const int ck = NT - 1 - tc - k; // column index ci with col_of(ci) == k, times 3
        const int rk = NT - 1 - tr - k; // row index ri with row_of(ri) == k, times 2
        const bool own_col = ck >= 0 && ck % 3 == 0, own_row = rk >= 0 && rk % 2 == 0;
        if (M0 > 0 && k < M0)
        {
            // band step: this wavefront's tiles of the panel and the strip wait in their staging slots
            {
                const int u_i = row_of(lane & 15);
                unsigned long long live = __builtin_amdgcn_ballot_w64(lane < AR && own_col && u_i > k && ((colk >> (u_i & 31)) & 1u));
                while (live)
                {
                    const int i = row_of(__builtin_ctzll(live));
                    live &= live - 1;
                    const unsigned sa = fresh(stage_lane_cr) + (unsigned)i * GP_STAGE_SLOT;
                    v4f64 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        x = __builtin_amdgcn_mfma_f64_16x16x4f64(IU[4 * q + l4][l15], *(gp_lptr)(unsigned long long)(sa + (unsigned)q * (4u * 17u * 8u)), x, 0, 0, 0);
                    const unsigned pa = fresh(frag_lane) + (unsigned)i * 128u;
#pragma unroll
                    for (int g = 0; g < 4; g++)
                        *(gp_lptr)(unsigned long long)(pa + (unsigned)g * (4u * (unsigned)ldp * 8u)) = x[g];
                    store_tile(i, k, x);
                }
            }
            {
                const int u_j = col_of(lane & 7);
                unsigned long long live = __builtin_amdgcn_ballot_w64(lane < AC && own_row && u_j > k && ((smap[u_j < 0 ? 0 : u_j] >> k) & 1u));
                while (live)
                {
                    const int j = col_of(__builtin_ctzll(live));
                    live &= live - 1;
                    const unsigned sa = fresh(stage_lane_cr) + (unsigned)(16 + j) * GP_STAGE_SLOT; // ([row][column] read transposed)
                    v4f64 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        y = __builtin_amdgcn_mfma_f64_16x16x4f64(*(gp_lptr)(unsigned long long)(sa + (unsigned)q * (4u * 17u * 8u)), IL[l15][4 * q + l4], y, 0, 0, 0);
                    const unsigned ya = fresh(tstore_lane) + (unsigned)j * 128u;
#pragma unroll
                    for (int g = 0; g < 4; g++)
                        *(gp_lptr)(unsigned long long)(ya + 32u * (unsigned)g) = y[g];
                    store_tile(k, j, y);
                }
            }
        }
        {
            if (own_col && k >= M0)
            {
                const int cik = ck / 3;
#pragma unroll
                for (int ri = 0; ri < RR; ri++)
                {
                    const int i = row_of(ri);
                    if (i <= k || !((colk >> i) & 1u))
                        continue;
#pragma unroll
                    for (int ci = 0; ci < RC; ci++)
                        if (ci == cik)
                            R[ri][ci] = finish_panel(R[ri][ci], i, k);
                }
            }
            if (own_row && k >= M0)
            {
                const int rik = rk / 2;
#pragma unroll
                for (int ci = 0; ci < RC; ci++)
                {
                    const int j = col_of(ci);
                    if (j <= k || j < 0 || !((colmap[ci] >> k) & 1u))
                        continue;
#pragma unroll
                    for (int ri = 0; ri < RR; ri++)
                        if (ri == rik)
                            R[ri][ci] = finish_strip(R[ri][ci], k, j);
                }
            }
        }
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
                {
                    t = load_tile(kn, kn); // (a band tile: the panels before k are in its memory copy)
                    if (reaches)
                    {
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            t = __builtin_amdgcn_mfma_f64_16x16x4f64(strip_frag(q, kn), panel_frag(q, kn), t, 0, 0, DG_NEG_A);
                    }
                }
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
