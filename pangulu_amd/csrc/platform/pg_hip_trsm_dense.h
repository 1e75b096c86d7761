// pg_hip_trsm_dense.h -- TSTRF / GESSM on dense mirrors with the f64 matrix cores (R64; included after pg_hip_dense.h).
//
// The sparse solve (trsm_sparse_kernel) walks a row/column entry by entry: nb strictly sequential steps, each
// waiting on L2 -- about a millisecond for a well-filled 256 x 256 block, on the critical path of every level.
// Here the block is solved panel by panel (16 columns / rows at a time) on its dense mirror:
//     TSTRF  X U = B :  X_p = (B_p - sum_{q<p} X_q U_qp) inv(U_pp)        p = 0..nb/16-1, column panels
//     GESSM  L X = B :  X_p = inv(L_pp) (B_p - sum_{q<p} L_pq X_q)        row panels
// with every product on v_mfma_f64_16x16x4_f64 and only nb/16 dependent steps.  The diagonal block comes as a dense
// "LU image" (L strictly below, U on and above the diagonal, as GETRF leaves its dense work image) whose 16 x 16
// diagonal tiles have been replaced by their inverses (upper part: inv(U_pp); strictly lower part: inv(L_pp) without
// its unit diagonal) by diag_tile_inverse_kernel.  Inverting 16 x 16 tiles instead of substituting through them is
// what dense GPU TRSMs do; it costs at most a few ulps more than substitution on these tiny, pivot-clamped tiles.
//
// Accumulators are laid out so that they load/store 128-byte column segments AND feed the final tile multiply as its
// B operand without a shuffle: TSTRF accumulates X^T tiles (register g of lane l = X(r0 + (l&15), 16p + (l>>4) + 4g)),
// GESSM accumulates X tiles (register g = X(16p + (l>>4) + 4g, c0 + (l&15))).
#pragma once

struct TrsmDenseTaskD
{
    double *b;        // mirror of the block being solved, overwritten by the solution
    const double *lu; // LU image of the diagonal block with inverted diagonal tiles
    u32 is_tstrf;
    u32 lu_map; // 1: the occupancy map behind `lu` describes the factorised block (written by getrf_tiled_f64_kernel / densify):
                // products with structurally empty factor tiles are skipped; 0: every tile of the factor counts as live
    const unsigned *progress; // chase launches only (getrf_trsm_chase_kernel): panels of `lu` the factorisation has published so far
};

// one wavefront per 16 x 16 diagonal tile: lane c < 16 computes column c of inv(U_pp) (back substitution) and of
// inv(L_pp) (forward substitution, unit diagonal), then the tile is overwritten.  T: 16 x 17 doubles of LDS owned by
// the calling wavefront; `sync` is a barrier for whoever shares the call (the wavefront's workgroup).
template <typename Sync>
__device__ inline void invert_diag_tile(double *D, int nb, int p, double (*T)[17], int lane, bool active, Sync sync)
{
    const size_t base = (size_t)(16 * p) * nb + 16 * p;
    if (active)
        for (int i = lane; i < 256; i += 64)
            T[i & 15][i >> 4] = D[base + (size_t)(i >> 4) * nb + (i & 15)]; // T[row][col]
    sync();
    double xu[16], xl[16];
    if (active && lane < 16)
    {
        const int c = lane;
        // U x = e_c  (upper, pivots clamped like the factorisation clamps them)
#pragma unroll
        for (int r = 15; r >= 0; r--)
        {
            double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 15; k > r; k--)
                s -= T[r][k] * xu[k];
            double piv = T[r][r];
            if ((piv < 0 ? -piv : piv) < PANGULU_TOL)
                piv = PANGULU_TOL;
            xu[r] = (r > c) ? 0.0 : s / piv;
        }
        // L x = e_c  (unit lower)
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < r; k++)
                s -= T[r][k] * xl[k];
            xl[r] = (r < c) ? 0.0 : s;
        }
    }
    sync();
    if (active && lane < 16)
    {
        const int c = lane;
#pragma unroll
        for (int r = 0; r < 16; r++)
            D[base + (size_t)c * nb + r] = (r <= c) ? xu[r] : xl[r];
    }
}

// images whose tiles were not inverted by the kernel that produced them (LU images rebuilt from received halves)
__global__ __launch_bounds__(64) void diag_tile_inverse_kernel(double *const *__restrict__ images, int nb)
{
    __shared__ double T[16][17];
    double *D = images[blockIdx.x / (nb / 16)];
    invert_diag_tile(D, nb, (int)(blockIdx.x % (nb / 16)), T, (int)threadIdx.x, true, []()
                     { __syncthreads(); });
}

// LU image of a diagonal block factorised on ANOTHER rank, from whichever of its halves have arrived (strictly lower CSC
// half, upper CSR half): one workgroup per image clears it and scatters the halves flat over their entries; the
// 16 x 16 diagonal tiles are then inverted by diag_tile_inverse_kernel like those of local images.  A missing half
// leaves its triangle zero; the solves that need only the other triangle never read it.
struct HalfImageJobD
{
    const u32 *lcp; // lower half, CSC (nullptr: not here)
    const u16 *lri;
    const val_t *lval;
    const u32 *urp; // upper half, CSR (nullptr: not here)
    const u16 *uci;
    const val_t *uval;
    double *dense;
};

#if defined(PG_DENSE_PANELS) // (real types: images of diagonal blocks factorised on another rank)
__global__ __launch_bounds__(1024) void half_image_kernel(const HalfImageJobD *__restrict__ jobs, int nb)
{
    extern __shared__ u32 s_ptr[]; // nb + 1 entries
    const HalfImageJobD J = jobs[blockIdx.x];
    for (int i = threadIdx.x; i < nb * nb / 2; i += blockDim.x)
        reinterpret_cast<double2 *>(J.dense)[i] = make_double2(0.0, 0.0);
    if (J.lcp)
    {
        for (int i = threadIdx.x; i <= nb; i += blockDim.x)
            s_ptr[i] = ptr0(J.lcp, i);
        __syncthreads(); // (also orders the clearing before the scatter)
        const u32 n = s_ptr[nb];
        for (u32 p = threadIdx.x; p < n; p += blockDim.x)
            J.dense[(size_t)mirror_column_of(s_ptr, nb, p) * nb + J.lri[p]] = J.lval[p];
    }
    __syncthreads();
    if (J.urp)
    {
        for (int i = threadIdx.x; i <= nb; i += blockDim.x)
            s_ptr[i] = J.urp[i];
        __syncthreads();
        const u32 n = s_ptr[nb];
        for (u32 p = threadIdx.x; p < n; p += blockDim.x)
            J.dense[(size_t)J.uci[p] * nb + mirror_column_of(s_ptr, nb, p)] = J.uval[p];
    }
}

#endif

// grid = tasks * (nb / 64); workgroup = 4 wavefronts; every wavefront solves 16 rows (TSTRF) or 16 columns (GESSM) of
// the block on its own and keeps ALL its finished 16 x 16 solution tiles in registers (NP tiles x 4 f64): a finished
// tile in accumulator layout is exactly the B operand the later panels need (register g <-> k = 4g + (l >> 4)).
// The factor's panel (U(0:16p+16, panel p) for TSTRF, L(panel p, 0:16p+16) for GESSM) is the same for the four
// wavefronts: it is staged through LDS once per workgroup (the next panel is prefetched into registers while the
// current one is consumed), laid out so that the MFMA A-operand reads are conflict-free:
//     TSTRF  sT[c * 258 + row]   (c = column within the panel)     GESSM  sT[k * 16 + r]   (r = row within the panel)
template <int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NP >= 16 ? 2 : 3))) void trsm_dense_f64_kernel(const TrsmDenseTaskD *__restrict__ tasks, unsigned long long *dbg, const u32 *__restrict__ work)
{
    unsigned long long stamp_ = dbg ? __builtin_amdgcn_s_memtime() : 0;
#define TRSM_STAMP(slot)                                                   \
    if (dbg && threadIdx.x == 0 && (blockIdx.x & 63) == 0)                 \
    {                                                                      \
        unsigned long long now_ = __builtin_amdgcn_s_memtime();            \
        atomicAdd(&dbg[slot], now_ - stamp_);                              \
        stamp_ = now_;                                                     \
    }
    constexpr int nb = NP * 16;
    constexpr int LDT = nb + 2;
    __shared__ __align__(16) double sT[16 * LDT];
    const int slabs = nb / 64;
    // (task, slab) from the launch's work list: slabs without pattern entries are left out
    const u32 item = work[logical_block_id((unsigned)slabs)]; // neighbours in the list read the same factor image: same XCD
    const unsigned bid = (item >> 2) * (unsigned)slabs + (item & 3u);
    const TrsmDenseTaskD T = tasks[bid / slabs];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int o0 = (bid % slabs) * 64 + wave * 16; // this wavefront's 16 rows (TSTRF) / columns (GESSM)
    double *__restrict__ Bm = T.b;
    const double *__restrict__ LU = T.lu;
    const bool tstrf = T.is_tstrf != 0;
    v4f64 xs[NP];

    // staging map: thread -> 16-byte pieces.  TSTRF panel p: column c = tid >> 4 of the panel, rows 2*(tid & 15) + 32 i
    // (i < ceil((16p+16)/32)); GESSM panel p: column k = (tid >> 3) + 32 i of the factor, rows 2*(tid & 7) of the panel.
    constexpr int NPIECE = (nb + 31) / 32;
    double2 pre[NPIECE];
    const int t_c = tid >> 4, t_r = 2 * (tid & 15);
    const int g_k = tid >> 3, g_r = 2 * (tid & 7);

#define TRSM_PREFETCH(p_)                                                                                         \
    {                                                                                                             \
        _Pragma("unroll") for (int i = 0; i < NPIECE; i++)                                                        \
        {                                                                                                         \
            pre[i] = make_double2(0.0, 0.0);                                                                      \
            if (tstrf)                                                                                            \
            {                                                                                                     \
                const int row = t_r + 32 * i;                                                                     \
                if (row < 16 * (p_) + 16)                                                                         \
                    pre[i] = *reinterpret_cast<const double2 *>(LU + (size_t)(16 * (p_) + t_c) * nb + row);       \
            }                                                                                                     \
            else                                                                                                  \
            {                                                                                                     \
                const int k = g_k + 32 * i;                                                                       \
                if (k < 16 * (p_) + 16)                                                                           \
                    pre[i] = *reinterpret_cast<const double2 *>(LU + (size_t)k * nb + 16 * (p_) + g_r);           \
            }                                                                                                     \
        }                                                                                                         \
    }
#define TRSM_STAGE(p_)                                                                                            \
    {                                                                                                             \
        _Pragma("unroll") for (int i = 0; i < NPIECE; i++)                                                        \
        {                                                                                                         \
            if (tstrf)                                                                                            \
            {                                                                                                     \
                const int row = t_r + 32 * i;                                                                     \
                if (row < 16 * (p_) + 16)                                                                         \
                    *reinterpret_cast<double2 *>(&sT[t_c * LDT + row]) = pre[i];                                  \
            }                                                                                                     \
            else                                                                                                  \
            {                                                                                                     \
                const int k = g_k + 32 * i;                                                                       \
                if (k < 16 * (p_) + 16)                                                                           \
                    *reinterpret_cast<double2 *>(&sT[k * 16 + g_r]) = pre[i];                                     \
            }                                                                                                     \
        }                                                                                                         \
    }

    // Structural zeros (occupancy map behind the mirror, pg_hip_dense.h): bit q of lv[w] = the 16 x 16 tile q of
    // wavefront w's strip holds pattern entries (fill included, so the solution's tile can be non-zero).  A tile that
    // is not set is zero before and after the solve: it is neither loaded, multiplied nor stored.  The workgroup starts
    // at the first panel in which any of its strips has a tile and leaves if there is none.
    const unsigned short *map = mirror_map(Bm, nb);
    unsigned lv[4];
#pragma unroll
    for (int w = 0; w < 4; w++)
    {
        const int strip = ((bid % slabs) * 64 + w * 16) >> 4;
        unsigned m = 0;
        if (tstrf)
        {
            for (int c = 0; c < NP; c++)
                m |= (((unsigned)map[c] >> strip) & 1u) << c;
        }
        else
            m = map[strip];
        lv[w] = m;
    }
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const unsigned my_lv = swave == 0 ? lv[0] : swave == 1 ? lv[1] : swave == 2 ? lv[2] : lv[3];
    const unsigned wg_lv = lv[0] | lv[1] | lv[2] | lv[3];
    if (wg_lv == 0)
    {
        if (dbg && threadIdx.x == 0 && (blockIdx.x & 63) == 0)
            atomicAdd(&dbg[7], 1ull << 40); // (empty workgroups counted in the high bits of the last slot)
        return;
    }
    const int wg_first = __builtin_ctz(wg_lv);

    // tile p, register g of lane l  <->  TSTRF: X(o0 + l15, 16p + l4 + 4g)    GESSM: X(16p + l4 + 4g, o0 + l15)
#pragma unroll
    for (int p = 0; p < NP; p++)
#pragma unroll
        for (int g = 0; g < 4; g++)
            xs[p][g] = !((my_lv >> p) & 1u) ? 0.0
                                             : (tstrf ? Bm[(size_t)(16 * p + l4 + 4 * g) * nb + o0 + l15] : Bm[(size_t)(o0 + l15) * nb + 16 * p + l4 + 4 * g]);
    TRSM_STAMP(0)
    TRSM_PREFETCH(wg_first)
    TRSM_STAMP(1)
#pragma unroll
    for (int p = 0; p < NP; p++)
    {
        if (!((wg_lv >> p) & 1u))
            continue; // (workgroup-uniform) no strip of this workgroup has a tile in panel p: X_p = 0 for all of them
        __syncthreads(); // everyone is done with the previous panel's image
        TRSM_STAMP(2)
        TRSM_STAGE(p)
        TRSM_STAMP(3)
        __syncthreads();
        TRSM_STAMP(4)
        {
            // the next panel any strip needs goes in flight now
            const unsigned later = (p + 1 < NP) ? (wg_lv >> (p + 1)) : 0u;
            if (later)
            {
                const int np = p + 1 + __builtin_ctz(later);
                TRSM_PREFETCH(np)
            }
        }
        if (!((my_lv >> p) & 1u))
            continue; // (wavefront-uniform; no barrier below)
        // four independent accumulation chains (one per k-quarter of a tile): a single chain of up to 60 dependent
        // MFMAs would leave the matrix core idle for most of each instruction's latency
        v4f64 part[4];
        part[0] = xs[p];
        part[1] = part[2] = part[3] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < p; q++)
        {
            if (!((my_lv >> q) & 1u))
                continue; // X_q = 0
#pragma unroll
            for (int kq = 0; kq < 4; kq++)
            {
                // TSTRF: A'[i = c][k] = -U(16q + k, 16p + c);  GESSM: A[i = r][k] = -L(16p + r, 16q + k);  B operand = xs[q][kq]
                const int k = 16 * q + kq * 4 + l4;
                const double a = tstrf ? -sT[l15 * LDT + k] : -sT[k * 16 + l15];
                part[kq] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xs[q][kq], part[kq], 0, 0, 0);
            }
        }
        TRSM_STAMP(5)
        v4f64 acc = (part[0] + part[1]) + (part[2] + part[3]);
        // multiply by the inverted diagonal tile (upper part: inv(U_pp); strictly lower part: inv(L_pp), unit diagonal)
        v4f64 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kq = 0; kq < 4; kq++)
        {
            const int k = kq * 4 + l4; // TSTRF: A''[i = c'][k = c] = invU(c, c');  GESSM: A[i = r'][k = r] = invL(r', r)
            double a;
            if (tstrf)
                a = (k <= l15) ? sT[l15 * LDT + 16 * p + k] : 0.0;
            else
                a = (l15 > k) ? sT[(16 * p + k) * 16 + l15] : ((l15 == k) ? 1.0 : 0.0);
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[kq], x, 0, 0, 0);
        }
        xs[p] = x;
        TRSM_STAMP(6)
    }
    // The solution tiles are written only now.  A store per panel sits in the same vmcnt counter as the next panel's
    // prefetch loads; with loads and stores mixed the counter cannot be waited on partially, so every staging step
    // would also wait for the previous panel's stores to come back.
#pragma unroll
    for (int p = 0; p < NP; p++)
    {
        if (!((my_lv >> p) & 1u))
            continue;
#pragma unroll
        for (int g = 0; g < 4; g++)
        {
            if (tstrf)
                Bm[(size_t)(16 * p + l4 + 4 * g) * nb + o0 + l15] = xs[p][g];
            else
                Bm[(size_t)(o0 + l15) * nb + 16 * p + l4 + 4 * g] = xs[p][g];
        }
    }
    TRSM_STAMP(7)
#undef TRSM_PREFETCH
#undef TRSM_STAGE
#undef TRSM_STAMP
}

// -----------------------------------------------------------------------------------------------------------------
// Barrier-free variant.  In-kernel stamps of the LDS-staged kernel above on the bench matrix: 19 % of a workgroup's time
// is spent at the start-of-panel barrier (its four strips have different live tiles), 11 + 2 % staging the factor's
// panel, 39 % of the workgroups have nothing to do at all.  Here every wavefront solves its strip on its own: the
// factor tiles it needs -- only those that meet a live solution tile -- come straight from the LU image (L2-resident:
// all solves of a level read the same few images) into MFMA operand registers, two tiles ahead of the matrix cores;
// no LDS, no barriers, a wavefront without live tiles leaves at once.
//   TSTRF  A'[i = c][k] = U(16q + k, 16p + c)  = LU[(16p + c) nb + 16q + k]
//   GESSM  A [i = r][k] = L(16p + r, 16q + k)  = LU[(16q + k) nb + 16p + r]
// -----------------------------------------------------------------------------------------------------------------
// CHASE: the image is being factorised by another workgroup of the same launch (getrf_trsm_chase_kernel): panel p of the solve
// starts when the factorisation has published panel p (T.progress > p), and everything read from the image -- factor tiles,
// inverted diagonal tiles, occupancy map -- is read `sc1` (the consumer half of the hand-off in pg_hip_getrf_tiled.h)
template <int NP, bool CHASE>
__device__ __forceinline__ void trsm_dense_direct_body(const TrsmDenseTaskD &T, int slab, int wave, int lane)
{
    constexpr int nb = NP * 16;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int o0 = slab * 64 + wave * 16; // this wavefront's 16 rows (TSTRF) / columns (GESSM)
    auto wait_for_panels = [&](unsigned n)
    {
        if (!CHASE)
            return;
        unsigned spins = 0;
        while (__hip_atomic_load(T.progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n)
        {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 22)) // (seconds: a factorisation that never publishes must abort the launch, not hang the device)
                __builtin_trap();
        }
    };
    double *__restrict__ Bm = T.b;
    const double *__restrict__ LU = T.lu;
    const bool tstrf = T.is_tstrf != 0;
    const unsigned short *map = mirror_map(Bm, nb);
    unsigned my_lv = 0;
    {
        const int strip = o0 >> 4;
        if (tstrf)
        {
            for (int c = 0; c < NP; c++)
                my_lv |= (((unsigned)map[c] >> strip) & 1u) << c;
        }
        else
            my_lv = map[strip];
        my_lv = (unsigned)__builtin_amdgcn_readfirstlane((int)my_lv);
    }
    if (my_lv == 0)
        return;
    // which factor tiles (q, p), q < p, hold pattern entries: column p of U's tile grid (TSTRF) / row p of L's (GESSM).
    // Diagonal blocks of the leaf levels are sparse at tile granularity too; a product with an empty factor tile is an
    // exact no-op, so it is neither loaded nor multiplied.
    const unsigned short *fmap = mirror_map(LU, nb);
    const bool use_fmap = T.lu_map != 0;
    // (the map words as scalars: the bit tests below then cost no vector instructions)
    unsigned fm[NP];
    wait_for_panels(1u); // (the factorisation writes the map before its first panel)
#pragma unroll
    for (int c = 0; c < NP; c++)
    {
        if (CHASE)
        {
            const unsigned w2 = use_fmap ? __hip_atomic_load(reinterpret_cast<const unsigned *>(fmap) + (c >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
            fm[c] = (unsigned)__builtin_amdgcn_readfirstlane((int)((w2 >> (16 * (c & 1))) & 0xFFFFu));
        }
        else
            fm[c] = use_fmap ? (unsigned)__builtin_amdgcn_readfirstlane((int)fmap[c]) : 0xFFFFu;
    }
    // factor tile (q, p): the four k-quarters of this lane's A operand.  Addresses are scalar bases (tile, k-quarter) plus one
    // constant lane offset: an f64 MFMA holds the SIMD's vector ALU for its 64 cycles, so vector address arithmetic is paid
    // for in matrix-pipe time (pg_hip_dense.h, dg_scalar_base)
    typedef const char __attribute__((address_space(1))) *gbytes;
    typedef const double __attribute__((address_space(1))) *gdouble_c;
    const unsigned a_voff = (unsigned)(tstrf ? l15 * nb + l4 : l4 * nb + l15) * 8u;
#define TRSM_A_LOAD(dst_, q_, p_)                                                                                   \
    {                                                                                                               \
        const gbytes ab_ = (gbytes)LU + (tstrf ? (size_t)(16 * (p_)) * nb + 16 * (q_) : (size_t)(16 * (q_)) * nb + 16 * (p_)) * 8; \
        _Pragma("unroll") for (int kq_ = 0; kq_ < 4; kq_++)                                                         \
        {                                                                                                           \
            const gdouble_c src_ = (gdouble_c)(dg_scalar_base(ab_ + (tstrf ? (size_t)(4 * kq_) : (size_t)(4 * kq_) * nb) * 8) + dg_lane_offset(a_voff)); \
            (dst_)[kq_] = CHASE ? __hip_atomic_load(src_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src_;      \
        }                                                                                                           \
    }
    // element g of this lane in solution tile p (TSTRF: X^T tiles, GESSM: X tiles; see the header)
    const unsigned x_voff = (unsigned)(tstrf ? l4 * nb + l15 : l15 * nb + l4) * 8u;
#define TRSM_X(p_, g_)                                                                                              \
    (*(double __attribute__((address_space(1))) *)(dg_scalar_base((gbytes)Bm + (tstrf ? (size_t)(16 * (p_) + 4 * (g_)) * nb + o0     \
                                                                                         : (size_t)o0 * nb + 16 * (p_) + 4 * (g_)) * 8) + \
                                                   dg_lane_offset(x_voff)))
    v4f64 xs[NP];
#pragma unroll
    for (int p = 0; p < NP; p++)
#pragma unroll
        for (int g = 0; g < 4; g++)
            xs[p][g] = !((my_lv >> p) & 1u) ? 0.0 : TRSM_X(p, g);
    // live products of panel p: solution tile q live AND factor tile (q, p) live
#define TRSM_LIVE_Q(out_, p_)                                                       \
    {                                                                               \
        unsigned fl_ = 0;                                                           \
        if (tstrf)                                                                  \
            fl_ = fm[p_]; /* tile column p of the image: bit q = rows 16q.. */      \
        else                                                                        \
        {                                                                           \
            _Pragma("unroll") for (int q_ = 0; q_ < NP; q_++)                       \
                fl_ |= ((fm[q_] >> (p_)) & 1u) << q_; /* tile row p: bit q = columns 16q.. */ \
        }                                                                           \
        (out_) = my_lv & fl_;                                                       \
    }
    // Factor tiles in flight: TRSM_SETS register sets of two tiles each, used in turn by the stages of a panel (two products
    // per stage; the loops are fully unrolled, so the set index is static and nothing is copied), and the inverted diagonal
    // tile of the panel (two sets, by panel parity).  The first tiles of panel p+1 and its diagonal tile are requested at
    // the end of panel p, before its diagonal multiply: without that every panel began with a full round trip to L2 (16 per
    // wavefront, a third of a lone workgroup's 70 us).  -DTRSM_SETS=3 requests the tiles of stage s + 2 at stage s (24 more
    // registers): measured, no difference (shell(398) solves 9.28 against 9.27 ms summed, fem27(112) 69.7 against 69.5;
    // profiles/r03y_trsm_prefetch_ab.log) -- the solves are not bound by the depth of their own load queue.
#ifndef TRSM_SETS
#define TRSM_SETS 2
#endif
    double at[TRSM_SETS][2][4], adb[2][4];
    bool fetched = false; // the first loads of the coming panel have been issued by the one before
    int set0 = 0;         // set of the coming panel's first stage (static after unrolling)
#define TRSM_FIRST_LOADS(p_, lq_, set_)                                             \
    {                                                                               \
        TRSM_A_LOAD(adb[(p_) & 1], p_, p_) /* needed last */                        \
        if ((p_) > 0 && (((lq_) >> 0) & 1u))                                        \
            TRSM_A_LOAD(at[set_][0], 0, p_)                                         \
        if ((p_) > 1 && (((lq_) >> 1) & 1u))                                        \
            TRSM_A_LOAD(at[set_][1], 1, p_)                                         \
        if (TRSM_SETS > 2 && (p_) > 2 && (((lq_) >> 2) & 1u))                       \
            TRSM_A_LOAD(at[((set_) + 1) % TRSM_SETS][0], 2, p_)                     \
        if (TRSM_SETS > 2 && (p_) > 3 && (((lq_) >> 3) & 1u))                       \
            TRSM_A_LOAD(at[((set_) + 1) % TRSM_SETS][1], 3, p_)                     \
    }
#pragma unroll
    for (int p = 0; p < NP; p++)
    {
        const int st = set0;
        set0 = (set0 + (p + 1) / 2) % TRSM_SETS; // a panel has (p + 1) / 2 stages
        if (!((my_lv >> p) & 1u))
        {
            fetched = false;
            continue; // (wavefront-uniform)
        }
        unsigned lq;
        TRSM_LIVE_Q(lq, p)
        if (!fetched)
        {
            wait_for_panels((unsigned)p + 1u);
            TRSM_FIRST_LOADS(p, lq, st)
        }
        v4f64 part[4];
        part[0] = xs[p];
        part[1] = part[2] = part[3] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < p; q += 2)
        {
            const int cur = (st + (q >> 1)) % TRSM_SETS, nxt = (cur + TRSM_SETS - 1) % TRSM_SETS; // (the set of stage s + SETS - 1)
            constexpr int ahead = 2 * (TRSM_SETS - 1);
            if (q + ahead < p && ((lq >> (q + ahead)) & 1u))
                TRSM_A_LOAD(at[nxt][0], q + ahead, p)
            if (q + ahead + 1 < p && ((lq >> (q + ahead + 1)) & 1u))
                TRSM_A_LOAD(at[nxt][1], q + ahead + 1, p)
            if ((lq >> q) & 1u)
            {
#pragma unroll
                for (int kq = 0; kq < 4; kq++)
                    part[kq] = __builtin_amdgcn_mfma_f64_16x16x4f64(at[cur][0][kq], xs[q][kq], part[kq], 0, 0, 1); // (NEG field: part -= a x)
            }
            if (q + 1 < p && ((lq >> (q + 1)) & 1u))
            {
#pragma unroll
                for (int kq = 0; kq < 4; kq++)
                    part[kq] = __builtin_amdgcn_mfma_f64_16x16x4f64(at[cur][1][kq], xs[q + 1][kq], part[kq], 0, 0, 1);
            }
        }
        fetched = false;
        if (p + 1 < NP && ((my_lv >> (p + 1)) & 1u))
        {
            unsigned lqn;
            TRSM_LIVE_Q(lqn, p + 1)
            wait_for_panels((unsigned)p + 2u);
            TRSM_FIRST_LOADS(p + 1, lqn, set0)
            fetched = true;
        }
        v4f64 acc = (part[0] + part[1]) + (part[2] + part[3]);
        // multiply by the inverted diagonal tile (upper part: inv(U_pp); strictly lower part: inv(L_pp), unit diagonal)
        v4f64 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kq = 0; kq < 4; kq++)
        {
            const int k = kq * 4 + l4; // TSTRF: A''[i = c'][k = c] = invU(c, c');  GESSM: A[i = r'][k = r] = invL(r', r)
            const double adk = adb[p & 1][kq];
            double a;
            if (tstrf)
                a = (k <= l15) ? adk : 0.0;
            else
                a = (l15 > k) ? adk : ((l15 == k) ? 1.0 : 0.0);
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[kq], x, 0, 0, 0);
        }
        xs[p] = x;
    }
#undef TRSM_FIRST_LOADS
#undef TRSM_LIVE_Q
#undef TRSM_A_LOAD
#pragma unroll
    for (int p = 0; p < NP; p++)
    {
        if (!((my_lv >> p) & 1u))
            continue;
#pragma unroll
        for (int g = 0; g < 4; g++)
            TRSM_X(p, g) = xs[p][g];
    }
#undef TRSM_X
}

template <int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NP >= 16 ? 2 : 3))) void trsm_dense_direct_f64_kernel(const TrsmDenseTaskD *__restrict__ tasks, const u32 *__restrict__ work)
{
    constexpr int slabs = NP * 16 / 64;
    // (task, slab) from the launch's work list: slabs without pattern entries are left out
    const u32 item = work[logical_block_id((unsigned)slabs)];
    const TrsmDenseTaskD T = tasks[item >> 2];
    trsm_dense_direct_body<NP, false>(T, (int)(item & 3u), __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), (int)(threadIdx.x & 63));
}

// structural flops of the solves that ran on the dense path (src/pangulu_kernel_interface.c:84-159): one workgroup
// per task.  T carries the CSC view of the solved block in vptr/vidx and the factor's pointer array in tptr.
//   TSTRF: every entry (r, c) costs 1 division + 2 per entry of U's row c right of the diagonal
//   GESSM: every entry (r, c) costs 2 per entry of L's column r
__global__ __launch_bounds__(256) void trsm_flop_count_kernel(const TrsmTaskD *__restrict__ tasks, int nb,
                                                              unsigned long long *flop_tstrf, unsigned long long *flop_gessm)
{
    const TrsmTaskD T = tasks[blockIdx.x];
    unsigned long long s = 0;
    if (T.is_tstrf)
    {
        for (int c = threadIdx.x; c < nb; c += blockDim.x)
        {
            const unsigned long long nbc = T.vptr[c + 1] - ptr0(T.vptr, c);
            const unsigned long long nu = T.tptr[c + 1] - T.tptr[c];
            if (nbc && nu)
                s += nbc * (1ull + 2ull * (nu - 1));
        }
    }
    else
    {
        const u32 nnz = T.vptr[nb];
        for (u32 p = threadIdx.x; p < nnz; p += blockDim.x)
        {
            const u32 r = T.vidx[p];
            s += 2ull * (T.tptr[r + 1] - T.tptr[r]);
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0 && s)
        atomicAdd(T.is_tstrf ? flop_tstrf : flop_gessm, s);
}
