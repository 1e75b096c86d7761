// pg_hip_dense.h -- dense-mode blocks: column-major mirrors in HBM and the f64 matrix-core SSSSM kernel.
// (included by pg_hip_platform.hip after the descriptor structs; R64 and CR64)
//
// CR64 (round 2): a complex block's mirror is TWO real planes (real part, imaginary part), each laid out exactly like an R64
// mirror (nb x nb doubles, occupancy map, saved-tiles slack), PG_PLANE_STRIDE(nb) doubles apart.  A complex update
// C -= A B then is four real ones on the planes -- C_re -= A_re B_re, C_re += A_im B_im, C_im -= A_re B_im, C_im -= A_im B_re --
// all through the real kernel below (a task carries a sign for its A operand; the two products of a destination plane
// are consecutive tasks of one group, so they share the accumulators and the destination pass).  Complex flops and real
// matrix-core flops are in the ratio 8 : 8, i.e. the complex rate equals the real MFMA rate.  The reference's GPU path
// calls cublasZgemm on densified blocks for this (...0201000.cu:778-816).
//
// North star: "MFMA applied only on blocks whose fill makes the update effectively a dense contraction".  The
// reference's rule is all-or-nothing (all three blocks completely full -> cuBLAS on the value arrays,
// ...0201000.cu:827-852; diagonal destination -> densify all three per task, :754-823).  Here a block whose fill
// reaches the dense threshold gets a persistent dense MIRROR (nb x nb doubles, zero outside its pattern) in a
// pool sized for 288 GB of HBM:
//   * updates INTO such a block accumulate in the mirror (MFMA kernel when both operands have mirrors, LDS kernel
//     with a dense destination column otherwise); its sparse values are refreshed once, right before its panel
//     operation (or it is handed to GETRF as the dense image directly);
//   * a finished L/U block used as an operand is densified once and then serves every later update.
// The sparse record stays the authoritative form of every finished block (exchange, solve, download).
#pragma once

// Occupancy map of a mirror (nb <= 256): 16 x u16 right behind the nb*nb values; word c = which 16-row slabs of the
// 16-column slab c hold pattern entries.  The pattern is symbolic (fill included) and never changes while the mirror
// belongs to the block, so whatever lands in the mirror later (updates, solves) stays inside the map.  The MFMA update
// skips K-slabs in which either operand is structurally zero, the dense solves skip empty strips and leading panels.
#define MIRROR_MAP_BYTES 64
// doubles from one plane of a mirror to the next (values + map + the slack a blocked GETRF saves diagonal tiles in)
__host__ __device__ inline size_t mirror_plane_stride(int nb)
{
    return (size_t)nb * nb + MIRROR_MAP_BYTES / sizeof(double) + (size_t)16 * nb;
}
__device__ inline const unsigned short *mirror_map(const double *mirror, int nb)
{
    return reinterpret_cast<const unsigned short *>(mirror + (size_t)nb * nb);
}

// ---------------------------------------------------------------------------------------------------------------
// C(nb x nb) -= sum_t A_t * B_t on dense mirrors.  Workgroup = 4 wavefronts = one 128 x 128 tile of C; each
// wavefront a 64 x 64 sub-tile as 4 x 4 accumulators of v_mfma_f64_16x16x4_f64.  K runs over all tasks of the
// group in steps of 16: the 128 x 16 slab of A and the 16 x 128 slab of B go through LDS ([k][m] / [k][n], row
// stride 144 doubles = conflict-free b64 fragment reads), the next slab is prefetched into registers while the
// current one feeds the matrix cores.  The product is formed transposed (A operand from the B slab, B operand from
// the A slab) so that accumulator register g of lane l is C(m0 + (l & 15), n0 + (l >> 4) + 4 g): the final
// read-modify-write of C moves 128-byte segments.
// ---------------------------------------------------------------------------------------------------------------
#define DG_TILE 128
#define DG_K 16
#ifndef DG_LD
// padded slab row (doubles).  144 (round 1) makes the MFMA fragment reads conflict-free but lands rows two apart on the same
// banks, which is exactly what the B-slab staging writes (eight threads write rows 0, 2, .. 14 of one column at once: an
// 8-way conflict per store); 146 / 148 spread those too: 24.6 -> 23.6 ms of update-kernel time (152: 24.8, 136: 24.2)
#define DG_LD 148
#endif
#define DG_WINDOW 16 // tasks whose bookkeeping is held in LDS at a time

// Wavefronts per workgroup.  4 (default): 64 x 64 sub-tiles (4 x 4 accumulators, 207 registers), two wavefronts per SIMD.
// 8: every wavefront owns a 64 x 32 sub-tile (2 x 4 accumulators, 123 registers), two workgroups = sixteen wavefronts per
// CU, four instruction streams per matrix-core pipe to cover each other's LDS waits, slab bookkeeping and barriers --
// measured equal (bench matrix 46.1-48.1 vs 46.2-47.4 ms per factorisation; poisson3d(64): 36.9 vs 37.2 ms of update-kernel
// time), so the variant with half the LDS operand reads per flop stays.
#ifndef DG_WAVES
#define DG_WAVES 4
#endif
#define DG_THREADS (64 * DG_WAVES)
#define DG_NI (DG_WAVES == 8 ? 2 : 4) // 16-column pieces of C per wavefront
#define DG_WAVES_PER_EU (DG_WAVES / 2)
__global__ __launch_bounds__(DG_THREADS) __attribute__((amdgpu_waves_per_eu(DG_WAVES_PER_EU, DG_WAVES_PER_EU))) void ssssm_dense_f64_kernel(const SsssmGroupD *__restrict__ groups,
                                                               const SsssmTaskD *__restrict__ tasks, int nb,
                                                               unsigned long long *__restrict__ product_counter,
                                                               unsigned long long *dbg, const u32 *__restrict__ work)
{
    // (debug stamps: every 64th workgroup adds its phase times; PANGULU_HIP_DEBUG_SSSSM)
    const bool stamping = dbg && threadIdx.x == 0 && (blockIdx.x & 63) == 0;
    unsigned long long stamp_ = dbg ? __builtin_amdgcn_s_memtime() : 0;
#define DG_STAMP(slot)                                                     \
    if (stamping)                                                          \
    {                                                                      \
        unsigned long long now_ = __builtin_amdgcn_s_memtime();            \
        atomicAdd(&dbg[slot], now_ - stamp_);                              \
        stamp_ = now_;                                                     \
    }
    // two images of each slab: while the matrix cores consume one, the next slab (already in registers) is written into
    // the other -- ONE barrier per slab instead of two (73.7 KB per workgroup: two workgroups still share a CU)
    __shared__ __align__(16) double sAb[2][DG_K * DG_LD];
    __shared__ __align__(16) double sBb[2][DG_K * DG_LD];
    const int tiles = nb / DG_TILE;
    const unsigned bid = logical_block_id((unsigned)(tiles * tiles)); // the tiles of one destination share operand halves: same XCD, same L2
    // (group, tile) of this workgroup: from the launch's work list (tiles no update of the group can reach are left out)
    const u32 item = work[bid];
    const int g = (int)(item >> 2);
    const int tile = (int)(item & 3u);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // (scalar: the skip tests below must be scalar branches)
    const int M0 = (tile % tiles) * DG_TILE, N0 = (tile / tiles) * DG_TILE; // workgroup tile origin
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * (16 * DG_NI);         // wavefront sub-tile inside it
    const int l15 = lane & 15, l4 = lane >> 4;
    const SsssmGroupD G = groups[g];

    v4f64 acc[DG_NI][4]; // [ni][mi]
#pragma unroll
    for (int ni = 0; ni < DG_NI; ni++)
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
            acc[ni][mi] = (v4f64){0.0, 0.0, 0.0, 0.0};

    // staging maps.  A slab (128 rows x 16 k, column-major source): thread -> rows 2*(tid & 63), +1 of k = (tid >> 6) + 4 i.
    // B slab (16 k x 128 cols): thread -> k pair 2*(tid & 7) of column (tid >> 3) + 32 i: 8 consecutive threads read one
    // 128-byte run of a column.
    // (with 8 wavefronts: A rows 2*(tid & 63), +1 of k = (tid >> 6) + 8 i, i < 2;  B k pair 2*(tid & 7) of column (tid >> 3) + 64 i)
    constexpr int NST = 16 / DG_WAVES;          // pieces each thread stages per operand (4 or 2)
    constexpr int A_KSTEP = DG_WAVES;           // k distance between a thread's A pieces
    constexpr int B_NSTEP = 8 * DG_WAVES;       // column distance between a thread's B pieces (32 or 64)
    const int a_m = 2 * (tid & 63), a_k = tid >> 6;
    const int b_k = 2 * (tid & 7), b_n = tid >> 3;
    const int a_slab = (tid & 63) >> 3;                            // which 16-row slab of the tile this thread stages
    const int b_slab = __builtin_amdgcn_readfirstlane(tid >> 7);   // 16-column slab of piece i: b_slab + (B_NSTEP / 16) i
    double2 ra[NST], rb[NST]; // next slab, in flight while the current one is consumed
#pragma unroll
    for (int i = 0; i < NST; i++)
        ra[i] = rb[i] = make_double2(0.0, 0.0);

    const u32 ntask = G.task_end - G.task_begin;
    const int steps_per_task = nb / DG_K;
    // Structural zeros, from the occupancy maps behind the mirrors (nb <= 256; otherwise everything counts as live).
    // For K-slab s of a task:  abits = which of the tile's eight 16-row slabs of A(:, s) hold pattern entries,
    //                          bbits = which of its eight 16-column slabs of B(s, :) do.
    // A slab is visited if both are non-zero; only live 16 x 16 pieces are fetched, and a wavefront issues the MFMAs
    // of a 16 x 16 x 16 product only when both its pieces are live -- for fill-in patterns of a few percent to a few
    // tens of percent that is a small fraction of the full contraction.  All of this is scalar (workgroup- or
    // wavefront-uniform) control flow.
    const bool mapped = nb <= 256;
    int cur_t = -1, nxt_step = -1;
    unsigned long long todo = 0;
    int done_steps = 0; // (only nb > 256: slabs of the current task handed out so far, 64 at a time)
    unsigned nxt_ab = 0xFF, nxt_bb = 0xFF, cur_ab, cur_bb, touched = 0, nprod = 0;
    const double *nxt_pa = nullptr, *nxt_pb = nullptr; // mirrors of the task the next step belongs to
    // Per-task bookkeeping of up to DG_WINDOW tasks at a time lives in LDS, filled by all 256 threads at once (thread
    // = one (task, slab) pair): chasing task -> mirror -> map through global memory once per task costs microseconds
    // of exposed latency per task, more than the slabs of a sparse update themselves.
    __shared__ unsigned short s_abbb[DG_WINDOW * 16];     // (bbits << 8) | abits of (task, slab); 0 = nothing to do
    __shared__ unsigned s_live[DG_WINDOW];               // per task: which slabs are live
    __shared__ const double *s_pa[DG_WINDOW], *s_pb[DG_WINDOW];
#if PG_PLANES > 1
    __shared__ double s_sign[DG_WINDOW]; // sign of the task's product (complex updates as four real ones)
    double nxt_sign = 1.0;
#endif
    int win0 = -DG_WINDOW; // first task of the window in the tables

#if PG_PLANES > 1
#define DG_SET_SIGN(t_, Tm_) s_sign[t_] = (Tm_).sign;
#define DG_GET_SIGN(i_) nxt_sign = s_sign[i_];
#else
#define DG_SET_SIGN(t_, Tm_)
#define DG_GET_SIGN(i_)
#endif
#define DG_FILL_WINDOW()                                                                             \
    {                                                                                                \
        __syncthreads(); /* nobody reads the previous window any more */                             \
        const int t_ = tid >> 4, s_ = tid & 15;                                                      \
        unsigned v_ = 0;                                                                             \
        if (tid < 256 && win0 + t_ < (int)ntask && s_ < steps_per_task)                              \
        {                                                                                            \
            const SsssmTaskD &Tm_ = tasks[G.task_begin + win0 + t_];                                 \
            const double *pa_ = reinterpret_cast<const double *>(Tm_.a.val), *pb_ = reinterpret_cast<const double *>(Tm_.b.val); \
            const unsigned ab_ = ((unsigned)mirror_map(pa_, nb)[s_] >> (M0 / 16)) & 0xFFu;           \
            const uint4 mb_ = *reinterpret_cast<const uint4 *>(mirror_map(pb_, nb) + N0 / 16);       \
            const unsigned w_[4] = {mb_.x, mb_.y, mb_.z, mb_.w};                                     \
            unsigned bb_ = 0;                                                                        \
            _Pragma("unroll") for (int c_ = 0; c_ < 8; c_++)                                         \
                bb_ |= (((w_[c_ >> 1] >> (16 * (c_ & 1))) >> s_) & 1u) << c_;                        \
            if (ab_ && bb_ && (!G.slab_mask || ((G.slab_mask >> s_) & 1u)))                          \
                v_ = (bb_ << 8) | ab_;                                                               \
            if (s_ == 0)                                                                             \
            {                                                                                        \
                s_pa[t_] = pa_;                                                                      \
                s_pb[t_] = pb_;                                                                      \
                DG_SET_SIGN(t_, Tm_)                                                                 \
            }                                                                                        \
        }                                                                                            \
        if (tid < 256)                                                                               \
            s_abbb[tid] = (unsigned short)v_;                                                        \
        const unsigned long long bal_ = __ballot(v_ != 0);                                           \
        if (tid < 256 && (tid & 63) < 4)                                                             \
            s_live[(tid >> 6) * 4 + (tid & 63)] = (unsigned)((bal_ >> (16 * (tid & 63))) & 0xFFFFull); \
        __syncthreads();                                                                             \
    }

#define DG_NEXT_STEP(out_)                                                                           \
    {                                                                                                \
        (out_) = -1;                                                                                 \
        while (mapped)                                                                               \
        {                                                                                            \
            if (todo)                                                                                \
            {                                                                                        \
                const int s_ = __builtin_ctzll(todo);                                                \
                todo &= todo - 1;                                                                    \
                const unsigned v_ = s_abbb[(cur_t - win0) * 16 + s_];                                \
                nxt_ab = v_ & 0xFFu;                                                                 \
                nxt_bb = v_ >> 8;                                                                    \
                nxt_pa = s_pa[cur_t - win0];                                                         \
                nxt_pb = s_pb[cur_t - win0];                                                         \
                DG_GET_SIGN(cur_t - win0)                                                            \
                (out_) = cur_t * steps_per_task + s_;                                                \
                break;                                                                               \
            }                                                                                        \
            if (++cur_t >= (int)ntask)                                                               \
                break;                                                                               \
            if (cur_t >= win0 + DG_WINDOW)                                                           \
            {                                                                                        \
                win0 = cur_t;                                                                        \
                DG_FILL_WINDOW()                                                                     \
            }                                                                                        \
            todo = s_live[cur_t - win0];                                                             \
        }                                                                                            \
        while (!mapped)                                                                              \
        {                                                                                            \
            if (todo)                                                                                \
            {                                                                                        \
                const int s_ = __builtin_ctzll(todo);                                                \
                todo &= todo - 1;                                                                    \
                (out_) = cur_t * steps_per_task + done_steps + s_;                                   \
                nxt_pa = reinterpret_cast<const double *>(tasks[G.task_begin + cur_t].a.val);        \
                nxt_pb = reinterpret_cast<const double *>(tasks[G.task_begin + cur_t].b.val);        \
                break;                                                                               \
            }                                                                                        \
            if (cur_t >= 0 && done_steps + 64 < steps_per_task)                                      \
            {                                                                                        \
                done_steps += 64;                                                                    \
                const int left_ = steps_per_task - done_steps;                                       \
                todo = left_ >= 64 ? ~0ull : ((1ull << left_) - 1ull);                               \
                continue;                                                                            \
            }                                                                                        \
            if (++cur_t >= (int)ntask)                                                               \
                break;                                                                               \
            done_steps = 0;                                                                          \
            todo = steps_per_task >= 64 ? ~0ull : ((1ull << steps_per_task) - 1ull);                 \
        }                                                                                            \
    }

#if PG_PLANES > 1
#define DG_APPLY_SIGN(v_) { (v_).x *= nxt_sign; (v_).y *= nxt_sign; }
#else
#define DG_APPLY_SIGN(v_)
#endif
#define DG_LOAD_SLAB(step_, ab_, bb_)                                                                \
    {                                                                                                \
        const int k0_ = ((step_) % steps_per_task) * DG_K;                                           \
        const double *A_ = nxt_pa + (size_t)(k0_ + a_k) * nb + M0 + a_m;                             \
        const double *B_ = nxt_pb + (size_t)(N0 + b_n) * nb + k0_ + b_k;                             \
        if (((ab_) >> a_slab) & 1u)                                                                  \
        {                                                                                            \
            _Pragma("unroll") for (int i_ = 0; i_ < NST; i_++)                                       \
            {                                                                                        \
                ra[i_] = *reinterpret_cast<const double2 *>(A_ + (size_t)(A_KSTEP * i_) * nb);       \
                DG_APPLY_SIGN(ra[i_])                                                                \
            }                                                                                        \
        }                                                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < NST; i_++)                                           \
            if (((bb_) >> (b_slab + (B_NSTEP / 16) * i_)) & 1u)                                      \
                rb[i_] = *reinterpret_cast<const double2 *>(B_ + (size_t)(B_NSTEP * i_) * nb);       \
    }

    DG_NEXT_STEP(nxt_step)
    DG_STAMP(0)
    if (nxt_step < 0)
    {
        if (stamping)
            atomicAdd(&dbg[7], 1ull << 32); // (empty workgroups in the high word)
        return; // nothing of these updates reaches this tile
    }
    DG_LOAD_SLAB(nxt_step, nxt_ab, nxt_bb)
    // A group that owns its destination and whose whole queue fits the bookkeeping window knows from the maps which
    // 16 x 16 pieces of C it is going to touch: their values go into the accumulators now (negated: acc = -C + sum A B,
    // C = -acc at the end), in flight together with the first slab, and the epilogue is stores only -- read-modify-write
    // at the end costs memory round trips that nothing hides.
    unsigned pre = 0;
    double *__restrict__ C = reinterpret_cast<double *>(G.cdense); // (CR64: one plane of the destination's mirror)
    if (mapped && ntask <= DG_WINDOW && !G.atomic)
    {
        unsigned m = 0;
        for (int e = lane; e < (int)ntask * 16; e += 64)
        {
            const unsigned v = s_abbb[e];
            const unsigned a4 = ((v & 0xFFu) >> (wm / 16)) & 0xFu, b4 = ((v >> 8) >> (wn / 16)) & ((1u << DG_NI) - 1u);
#pragma unroll
            for (int ni = 0; ni < DG_NI; ni++)
                if ((b4 >> ni) & 1u)
                    m |= a4 << (4 * ni);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            m |= (unsigned)__shfl_xor((int)m, off, 64);
        pre = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
#pragma unroll
        for (int ni = 0; ni < DG_NI; ni++)
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
                if ((pre >> (4 * ni + mi)) & 1u)
                {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        acc[ni][mi][r] = -C[(size_t)(N0 + wn + ni * 16 + l4 + 4 * r) * nb + (M0 + wm + mi * 16 + l15)];
                }
    }
#define DG_STORE_LDS(buf_)                                                                           \
    {                                                                                                \
        double *sA_ = sAb[buf_], *sB_ = sBb[buf_];                                                   \
        /* (pieces that were not fetched hold stale finite values; no MFMA reads them) */            \
        _Pragma("unroll") for (int i_ = 0; i_ < NST; i_++)                                           \
        {                                                                                            \
            *reinterpret_cast<double2 *>(&sA_[(a_k + A_KSTEP * i_) * DG_LD + a_m]) = ra[i_];         \
            sB_[b_k * DG_LD + b_n + B_NSTEP * i_] = rb[i_].x;                                        \
            sB_[(b_k + 1) * DG_LD + b_n + B_NSTEP * i_] = rb[i_].y;                                  \
        }                                                                                            \
    }
    // software pipeline: slab s in LDS image `buf` feeds the matrix cores, slab s+1 sits in registers (its loads went out one
    // iteration ago) and goes into the other image afterwards, the loads of slab s+2 go out, one barrier
    cur_ab = nxt_ab;
    cur_bb = nxt_bb;
    DG_STAMP(1)
    DG_STORE_LDS(0)
    DG_STAMP(2)
    DG_NEXT_STEP(nxt_step)
    if (nxt_step >= 0)
        DG_LOAD_SLAB(nxt_step, nxt_ab, nxt_bb)
    __syncthreads();
    DG_STAMP(3)
    int buf = 0;
    for (;;)
    {
        const double *sA = sAb[buf], *sB = sBb[buf];
        const unsigned a4 = (cur_ab >> (wm / 16)) & 0xFu, b4 = (cur_bb >> (wn / 16)) & ((1u << DG_NI) - 1u);
        if (a4 && b4)
        {
            nprod += (unsigned)(__builtin_popcount(a4) * __builtin_popcount(b4));
#pragma unroll
            for (int ni = 0; ni < DG_NI; ni++)
                if ((b4 >> ni) & 1u)
                    touched |= a4 << (4 * ni);
#pragma unroll
            for (int kq = 0; kq < DG_K / 4; kq++)
            {
                double fa[4], fb[DG_NI];
#pragma unroll
                for (int mi = 0; mi < 4; mi++)
                    fa[mi] = sA[(kq * 4 + l4) * DG_LD + wm + mi * 16 + l15];
#pragma unroll
                for (int ni = 0; ni < DG_NI; ni++)
                    fb[ni] = sB[(kq * 4 + l4) * DG_LD + wn + ni * 16 + l15];
#pragma unroll
                for (int ni = 0; ni < DG_NI; ni++)
                {
                    if (!((b4 >> ni) & 1u))
                        continue;
#pragma unroll
                    for (int mi = 0; mi < 4; mi++)
                        if ((a4 >> mi) & 1u)
                            acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[ni], fa[mi], acc[ni][mi], 0, 0, 0);
                }
            }
        }
        DG_STAMP(5)
        if (stamping)
            atomicAdd(&dbg[7], 1ull); // slab steps in the low word
        if (nxt_step < 0)
            break;
        cur_ab = nxt_ab;
        cur_bb = nxt_bb;
        DG_STORE_LDS(buf ^ 1) // (waits for the slab's loads: they have been in flight since before the products above)
        DG_STAMP(2)
        DG_NEXT_STEP(nxt_step)
        if (nxt_step >= 0)
            DG_LOAD_SLAB(nxt_step, nxt_ab, nxt_bb)
        DG_STAMP(4)
        __syncthreads(); // the other image is complete, and everyone is done reading this one
        DG_STAMP(3)
        buf ^= 1;
    }
#undef DG_STORE_LDS
#undef DG_NEXT_STEP
#undef DG_FILL_WINDOW
#undef DG_LOAD_SLAB
#undef DG_APPLY_SIGN
#undef DG_SET_SIGN
#undef DG_GET_SIGN
    if (product_counter && lane == 0 && nprod)
        atomicAdd(product_counter, (unsigned long long)nprod); // 16 x 16 x 16 products issued to the matrix cores

    if (pre)
    {
#pragma unroll
        for (int ni = 0; ni < DG_NI; ni++)
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
                if ((pre >> (4 * ni + mi)) & 1u)
                {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        C[(size_t)(N0 + wn + ni * 16 + l4 + 4 * r) * nb + (M0 + wm + mi * 16 + l15)] = -acc[ni][mi][r];
                }
        touched = 0; // (= pre: everything has been written)
    }
    // Read-modify-write of the touched 16 x 16 pieces, a column of four pieces (16 values per lane) at a time: all its
    // loads go out together, then all its stores.  Piece by piece (load 4, store 4, load 4 ...) every load waits for the
    // stores before it as well -- loads and stores share one counter and cannot be waited on separately -- which made this
    // epilogue 16 dependent memory round trips, 14 us of a workgroup's 42.
#pragma unroll
    for (int ni = 0; ni < DG_NI; ni++)
    {
        const unsigned t4 = (touched >> (4 * ni)) & 0xFu;
        if (!t4)
            continue; // no product reached this column of pieces
        if (G.atomic)
        {
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
            {
                if (!((t4 >> mi) & 1u))
                    continue;
#pragma unroll
                for (int r = 0; r < 4; r++)
                {
                    const size_t off = (size_t)(N0 + wn + ni * 16 + l4 + 4 * r) * nb + (M0 + wm + mi * 16 + l15);
                    if (acc[ni][mi][r] != 0.0)
                        atomicAdd(&C[off], -acc[ni][mi][r]);
                }
            }
            continue;
        }
        double old[4][4];
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                old[mi][r] = ((t4 >> mi) & 1u) ? C[(size_t)(N0 + wn + ni * 16 + l4 + 4 * r) * nb + (M0 + wm + mi * 16 + l15)] : 0.0;
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
        {
            if (!((t4 >> mi) & 1u))
                continue; // no product reached this 16 x 16 piece of C
#pragma unroll
            for (int r = 0; r < 4; r++)
                C[(size_t)(N0 + wn + ni * 16 + l4 + 4 * r) * nb + (M0 + wm + mi * 16 + l15)] = old[mi][r] - acc[ni][mi][r];
        }
    }
    DG_STAMP(6)
#undef DG_STAMP
}

// ---------------------------------------------------------------------------------------------------------------
// mirror maintenance.  One workgroup per block.
// ---------------------------------------------------------------------------------------------------------------
struct MirrorJobD
{
    BlkView lo;     // CSC view (off-diagonal block, or strictly-lower half of a diagonal block)
    // upper half of a diagonal block through its column view (DiagAux): column pointers, row indices, and the position
    // of every entry in the CSR value array `uval`; ucp == nullptr for off-diagonal blocks
    const u32 *ucp;
    const u16 *uri;
    const u32 *uvi;
    val_t *uval;
    double *dense;  // nb x nb column-major (CR64: the real plane; the imaginary one mirror_plane_stride(nb) doubles behind)
    const double *diag_tiles; // sparsify of a fresh LU image: its nb/16 diagonal tiles as GETRF left them (16 x 16 column-major
                              // each; the image's own have been inverted in place since), or nullptr
    unsigned long long move_bytes; // host side only (statistics): record entries read + image entries written, or the reverse
};

// Entries ptr[c0] .. ptr[c1] of a CSC block are one contiguous run: the workgroup walks it flat (coalesced, no
// per-column pointer chasing) and finds the column of entry p by bisection in an LDS copy of the pointer slice.
#define MIRROR_MAX_COLS 256
__device__ inline int mirror_column_of(const u32 *sp, int ncols, u32 p)
{
    int lo = 0, hi = ncols; // sp[lo] <= p < sp[hi]
    while (hi - lo > 1)
    {
        const int mid = (lo + hi) >> 1;
        if (sp[mid] <= p)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

__device__ inline void mirror_put(double *dense, size_t at, int nb, val_t v)
{
#if PG_PLANES > 1
    dense[at] = v.re;
    dense[at + mirror_plane_stride(nb)] = v.im;
#else
    (void)nb;
    dense[at] = v;
#endif
}
__device__ inline val_t mirror_get(const double *dense, size_t at, int nb)
{
#if PG_PLANES > 1
    return val_t{dense[at], dense[at + mirror_plane_stride(nb)]};
#else
    (void)nb;
    return dense[at];
#endif
}

// grid = (jobs, slices): a workgroup owns a run of columns (whole 16-column slabs) of one block, so that a launch with
// a single job (the tail of the elimination tree) still spreads over many CUs
__global__ __launch_bounds__(256) void densify_kernel(const MirrorJobD *__restrict__ jobs, int nb)
{
    __shared__ unsigned occ[16];
    __shared__ u32 sp[MIRROR_MAX_COLS + 1], su[MIRROR_MAX_COLS + 1];
    const MirrorJobD J = jobs[blockIdx.x];
    int per = (nb + (int)gridDim.y - 1) / (int)gridDim.y;
    if (per > MIRROR_MAX_COLS)
        per = MIRROR_MAX_COLS;
    const bool mapped = nb <= 256;
    unsigned short *map = reinterpret_cast<unsigned short *>(J.dense + (size_t)nb * nb);
    // (nb > MIRROR_MAX_COLS * slices: the workgroup takes several runs of columns, one after the other)
    for (int c0 = (int)blockIdx.y * per; c0 < nb; c0 += per * (int)gridDim.y)
    {
        const int c1 = min(nb, c0 + per), ncols = c1 - c0;
        __syncthreads();
        if (threadIdx.x < 16)
            occ[threadIdx.x] = 0;
        for (int i = threadIdx.x; i <= ncols; i += blockDim.x)
        {
            sp[i] = ptr0(J.lo.ptr, c0 + i);
            su[i] = J.ucp ? J.ucp[c0 + i] : 0u;
        }
        __syncthreads();
        const u32 e0 = sp[0], e1 = sp[ncols], f0 = su[0], f1 = su[ncols];
        const bool by_tiles = mapped && !J.ucp;
        if (by_tiles)
        {
            // Off-diagonal block: everything that reads the mirror goes by the occupancy map (the MFMA update and the
            // dense solves touch live 16 x 16 tiles only, sparsify reads pattern entries), so only live tiles are
            // cleared -- for fill of a few percent that is a fraction of the nb*nb image.  (Dead tiles keep whatever
            // the memory held; the sparse-update kernel carries such values through its column pass unchanged.)
            for (u32 p0 = e0 + threadIdx.x; p0 < e1; p0 += 4 * blockDim.x)
            {
                u32 r4[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                    r4[u] = p0 + u * blockDim.x < e1 ? J.lo.idx[p0 + u * blockDim.x] : 0u;
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (p0 + u * blockDim.x < e1)
                        atomicOr(&occ[(c0 + mirror_column_of(sp, ncols, p0 + u * blockDim.x)) >> 4], 1u << (r4[u] >> 4));
            }
            __syncthreads();
            const int tpc = nb / 2; // 16-byte pieces per column; eight of them per 16-row slab
            for (int i = threadIdx.x; i < ncols * tpc; i += blockDim.x)
            {
                const int c = c0 + i / tpc, within = i % tpc;
                if ((occ[c >> 4] >> (within >> 3)) & 1u)
                {
                    reinterpret_cast<double2 *>(J.dense + (size_t)c * nb)[within] = make_double2(0.0, 0.0);
#if PG_PLANES > 1
                    reinterpret_cast<double2 *>(J.dense + mirror_plane_stride(nb) + (size_t)c * nb)[within] = make_double2(0.0, 0.0);
#endif
                }
            }
        }
        else
        {
#if PG_PLANES > 1
          for (int pl_ = 0; pl_ < PG_PLANES; pl_++)
          {
            double *base = J.dense + pl_ * mirror_plane_stride(nb) + (size_t)c0 * nb;
#else
          {
            double *base = J.dense + (size_t)c0 * nb;
#endif
            const int words = ncols * nb;
            if ((words & 1) == 0 && (((size_t)c0 * nb) & 1) == 0)
            {
                double2 *d2 = reinterpret_cast<double2 *>(base); // mirrors are 16-byte aligned
                for (int i = threadIdx.x; i < words / 2; i += blockDim.x)
                    d2[i] = make_double2(0.0, 0.0);
            }
            else
                for (int i = threadIdx.x; i < words; i += blockDim.x)
                    base[i] = 0.0;
          }
        }
        __syncthreads();
        // four entries per thread and pass: their index and value loads go out together, then the four stores (one
        // entry at a time, every iteration's loads wait for the previous iteration's store as well: loads and stores
        // share one counter)
        for (u32 p0 = e0 + threadIdx.x; p0 < e1; p0 += 4 * blockDim.x)
        {
            u32 r4[4];
            val_t v4[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
                const u32 p = p0 + u * blockDim.x;
                r4[u] = p < e1 ? J.lo.idx[p] : 0u;
                v4[u] = p < e1 ? J.lo.val[p] : v_make(0);
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
                const u32 p = p0 + u * blockDim.x;
                if (p < e1)
                {
                    const int c = c0 + mirror_column_of(sp, ncols, p);
                    mirror_put(J.dense, (size_t)c * nb + r4[u], nb, v4[u]);
                    if (mapped && !by_tiles)
                        atomicOr(&occ[c >> 4], 1u << (r4[u] >> 4));
                }
            }
        }
        for (u32 p = f0 + threadIdx.x; p < f1; p += blockDim.x) // upper half of a diagonal block
        {
            const int c = c0 + mirror_column_of(su, ncols, p);
            const u32 r = J.uri[p];
            mirror_put(J.dense, (size_t)c * nb + r, nb, J.uval[J.uvi[p]]);
            if (mapped)
                atomicOr(&occ[c >> 4], 1u << (r >> 4));
        }
        if (mapped)
        {
            __syncthreads();
            const int s_ = threadIdx.x;
            if (s_ < 16 && s_ * 16 >= c0 && s_ * 16 < c1)
            {
                map[s_] = (unsigned short)occ[s_];
#if PG_PLANES > 1
                reinterpret_cast<unsigned short *>(J.dense + mirror_plane_stride(nb) + (size_t)nb * nb)[s_] = (unsigned short)occ[s_];
#endif
            }
        }
    }
}

__global__ __launch_bounds__(256) void sparsify_kernel(const MirrorJobD *__restrict__ jobs, int nb)
{
    __shared__ u32 sp[MIRROR_MAX_COLS + 1], su[MIRROR_MAX_COLS + 1];
    const MirrorJobD J = jobs[blockIdx.x];
    int per = (nb + (int)gridDim.y - 1) / (int)gridDim.y;
    if (per > MIRROR_MAX_COLS)
        per = MIRROR_MAX_COLS;
    for (int c0 = (int)blockIdx.y * per; c0 < nb; c0 += per * (int)gridDim.y)
    {
        const int c1 = min(nb, c0 + per), ncols = c1 - c0;
        __syncthreads();
        for (int i = threadIdx.x; i <= ncols; i += blockDim.x)
        {
            sp[i] = ptr0(J.lo.ptr, c0 + i);
            su[i] = J.ucp ? J.ucp[c0 + i] : 0u;
        }
        __syncthreads();
        const u32 e0 = sp[0], e1 = sp[ncols], f0 = su[0], f1 = su[ncols];
        for (u32 p0 = e0 + threadIdx.x; p0 < e1; p0 += 4 * blockDim.x) // (four entries per pass, see densify_kernel)
        {
            u32 r4[4];
            val_t v4[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
                r4[u] = p0 + u * blockDim.x < e1 ? J.lo.idx[p0 + u * blockDim.x] : 0u;
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
                const u32 p = p0 + u * blockDim.x;
                const int c = c0 + mirror_column_of(sp, ncols, p < e1 ? p : e0);
                const u32 r = r4[u];
#if PG_PLANES == 1
                v4[u] = p >= e1 ? 0.0
                                : (J.diag_tiles && (r >> 4) == (u32)(c >> 4)) ? J.diag_tiles[((c >> 4) << 8) + ((c & 15) << 4) + (r & 15)]
                                                                            : J.dense[(size_t)c * nb + r];
#else
                v4[u] = p >= e1 ? v_make(0) : mirror_get(J.dense, (size_t)c * nb + r, nb);
#endif
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (p0 + u * blockDim.x < e1)
                    J.lo.val[p0 + u * blockDim.x] = v4[u];
        }
        for (u32 p = f0 + threadIdx.x; p < f1; p += blockDim.x)
        {
            const int c = c0 + mirror_column_of(su, ncols, p);
            const u32 r = J.uri[p];
#if PG_PLANES == 1
            J.uval[J.uvi[p]] = (J.diag_tiles && (r >> 4) == (u32)(c >> 4)) ? J.diag_tiles[((c >> 4) << 8) + ((c & 15) << 4) + (r & 15)]
                                                                        : J.dense[(size_t)c * nb + r];
#else
            J.uval[J.uvi[p]] = mirror_get(J.dense, (size_t)c * nb + r, nb);
#endif
        }
    }
}

// structural flops of C -= A*B: 2 * sum over entries (k, j) of B of nnz(A(:, k))  (src/pangulu_kernel_interface.c:161-176)
__global__ __launch_bounds__(256) void ssssm_flop_count_kernel(const SsssmTaskD *__restrict__ tasks, int nb,
                                                               unsigned long long *flop_counter)
{
    const SsssmTaskD T = tasks[blockIdx.x];
    if (!T.count)
        return; // (CR64: the other three real products of the same complex update)
    const u32 nnzb = T.b.ptr[nb];
    unsigned long long s = 0;
    for (u32 p = threadIdx.x; p < nnzb; p += blockDim.x)
    {
        const u32 k = T.b.idx[p];
        s += T.a.ptr[k + 1] - ptr0(T.a.ptr, (int)k);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0 && s)
        atomicAdd(flop_counter, 2ull * s);
}
