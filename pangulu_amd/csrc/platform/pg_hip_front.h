// pg_hip_front.h -- the MFMA update kernel for DENSE FRONTS (R64; CR64 plane by plane): C(128 x 128 tile) -= sum_t A_t B_t where
// every 16 x 16 piece of the operands that meets the tile holds pattern entries.
// (included by pg_hip_platform.hip after pg_hip_dense.h; tools/microbench/front_gemm.hip times it stand-alone)
//
// Where it applies.  The reference's direct-gemm case (all three blocks completely full -> cuBLAS on the value arrays,
// ...0201000.cu:827-852) generalised to tiles: on 3D problems the top separators of the elimination tree are dense fronts and
// ARE the factorisation -- fem27(112): 81 % of the update kernel's time in launches of more than 16 000 workgroups whose
// slabs are all live.  The general kernel (pg_hip_dense.h) pays for what such tiles do not need: occupancy tests per
// (task, K-slab), a compacted step list per window of 16 tasks, conditional MFMAs, dummy loads for dead pieces -- and keeps
// ONE slab of operands in flight, staged through registers (phase stamps on fem27(80): 35 % of a workgroup's time in the
// matrix-core phase, 23 % waiting for the slab, 18 % at the two barriers, 15 % issuing the next loads).
//
// What it does instead.
//  * Operand slabs go from HBM/L2 straight into LDS (global_load_lds_dwordx4: 1 KiB per wave instruction, no staging
//    registers, no ds_write pass), STAGES slabs deep: with S stages, S - 1 slabs are in flight while the matrix cores work
//    on one; waits are counted (s_waitcnt vmcnt(4 (S - 2)): every wave issues exactly four DMA instructions per slab) and
//    the barrier is a bare s_barrier -- __syncthreads() would drain the DMA queue (cdna_hip_programming.md §5).
//  * An LDS-DMA writes 64 x 16 bytes CONTIGUOUSLY; what goes where is decided by the per-lane SOURCE address:
//      A slab (16 columns k of 128 rows, mirror column-major): one instruction per column = 1 KiB; columns 144 doubles
//      apart (k*16 + m covers all 32 double-banks for the two columns a half-wave reads);
//      B slab (16 rows k of 128 columns): one instruction per 8 columns, eight lanes per column fetching its 16 consecutive
//      k's (one 128-byte line) as pairs; lane j of column n fetches pair j ^ ((n >> 1) & 7) -- an XOR swizzle on the
//      source side -- so that the MFMA fragment reads (a half-wave = 16 columns x 2 rows of one pair) hit 32 different
//      double-banks.
//  * No bookkeeping: step = (task, slab) in order, operand pointers by scalar loads from the task descriptors.
//  * acc = -sum A B on the matrix cores' NEG field, C += acc at the end (plain read-modify-write, or floating-point
//    atomics for a split queue).
#pragma once

#define FR_TILE 128
#define FR_KS 16
#define FR_LDA 144                                        // doubles between two k columns of the A image
#define FR_STAGE_DOUBLES (FR_KS * FR_LDA + FR_TILE * FR_KS) // A image + B image of one slab: 34 816 bytes
#define FR_THREADS 512

typedef const char __attribute__((address_space(1))) *fr_gptr;
typedef void __attribute__((address_space(3))) *fr_lptr;

// STAGES: LDS stages of the operand pipeline.  PREFETCH: the MFMA fragments of k-quarter kq + 1 are read while the matrix
// cores work on kq (12 more registers).  `unit`: consecutive workgroups that go to the same XCD (logical_block_id): 4 =
// the tiles of one destination; 4 g = the tiles of g consecutive destinations, which share an operand in the scheduler's
// release order (all updates of one finished L block, or of one finished U block, are queued in a row).
template <int STAGES, bool PREFETCH>
__global__ __launch_bounds__(FR_THREADS, (STAGES <= 2 ? 4 : 2)) void ssssm_front_f64_kernel(const SsssmTaskD *__restrict__ tasks, int nb, const SsssmWorkD *__restrict__ work,
                                                                     unsigned long long *__restrict__ product_counter, unsigned unit)
{
    __shared__ __align__(16) double lds[STAGES * FR_STAGE_DOUBLES];
    const int tiles = nb / FR_TILE;
    const unsigned bid = logical_block_id(unit ? unit : (unsigned)(tiles * tiles));
    const SsssmWorkD G = work[bid];
    const int tile = (int)G.tile;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M0 = (tile % tiles) * FR_TILE, N0 = (tile / tiles) * FR_TILE;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 32; // wavefront sub-tile: 64 rows x 32 columns = 4 x 2 accumulators
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ntask = (int)(G.task_end - G.task_begin);
    const int steps_shift = nb == 256 ? 4 : 3; // nb / 16 slabs per task (dense mode: nb = 128 or 256)
    const int steps_per_task = 1 << steps_shift;
    const int T = ntask << steps_shift;
    const SsssmTaskD *my_tasks = tasks + G.task_begin;

    // per-lane source offsets of the DMA instructions (bytes; constant over the kernel)
    //   A: lane l fetches rows 2l, 2l+1 of column k  ->  + (k0 + k) * nb * 8 from the scalar side
    const unsigned a_voff = (unsigned)lane * 16u;
    //   B: lane l = 8 c + j fetches pair j ^ ((n >> 1) & 7) of column n = 8 g + c  (g = instruction); (n >> 1) & 7 = ((8 g + c) >> 1) & 7
    //      = (4 g + (c >> 1)) & 7: depends on g through 4 g & 7 = 4 (g & 1) only
    const int bc = lane >> 3, bj = lane & 7;
    unsigned b_voff[2]; // for even / odd g
#pragma unroll
    for (int par = 0; par < 2; par++)
        b_voff[par] = ((unsigned)bc * (unsigned)nb + 2u * (unsigned)(bj ^ ((4 * par + (bc >> 1)) & 7))) * 8u;

    auto issue = [&](int st)
    {
        // four DMA instructions per wave: A columns `wave`, `wave + 8`; B column groups `wave`, `wave + 8`
        const int t = st >> steps_shift, k0 = (st & (steps_per_task - 1)) * FR_KS;
        const fr_gptr pa = (fr_gptr)reinterpret_cast<const char *>(my_tasks[t].a.val);
        const fr_gptr pb = (fr_gptr)reinterpret_cast<const char *>(my_tasks[t].b.val);
        double *stage = lds + (st % STAGES) * FR_STAGE_DOUBLES;
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
            const int k = wave + 8 * h;
            const fr_gptr src = dg_scalar_base(pa + ((size_t)(k0 + k) * nb + M0) * 8) + a_voff;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src, (fr_lptr)(stage + k * FR_LDA), 16, 0, 0);
        }
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
            const int g = wave + 8 * h;
            const fr_gptr src = dg_scalar_base(pb + ((size_t)(N0 + 8 * g) * nb + k0) * 8) + b_voff[wave & 1]; // (g & 1 = wave & 1)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src, (fr_lptr)(stage + FR_KS * FR_LDA + g * 128), 16, 0, 0);
        }
    };

    // C += acc: accumulator register r of lane l is C(M0 + wm + mi*16 + (l & 15), N0 + wn + ni*16 + 4 r + (l >> 4))
    double __attribute__((address_space(1))) *C = (double __attribute__((address_space(1))) *)reinterpret_cast<double *>(G.cdense);
    const unsigned c_voff = ((unsigned)l4 * (unsigned)nb + (unsigned)l15) * 8u;
#define FR_C(ni_, mi_, r_)                                                                           \
    (*(double __attribute__((address_space(1))) *)(dg_scalar_base((const char __attribute__((address_space(1))) *)C +                    \
                                                                  ((size_t)(N0 + wn + (ni_) * 16 + 4 * (r_)) * nb + M0 + wm) * 8) + \
                                                   dg_lane_offset(c_voff) + (mi_) * 128))
    // A workgroup that owns its destination (no atomics) takes the tile into the accumulators up front (round 4; the general kernel
    // has done so since round 3): the loads are older than every DMA instruction, so the first step's wait covers them, they travel
    // beside the first slab, the matrix cores compute C - sum A B and the epilogue is stores only -- a read-modify-write at the end
    // is a dependent round trip nothing hides (stand-alone: FR_PRELOAD=0 for the old epilogue).
#ifndef FR_PRELOAD
#define FR_PRELOAD 1
#endif
    const bool preload = FR_PRELOAD && !G.atomic && T > 0;
    v4f64 acc[2][4]; // [ni][mi]
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
        {
            if (preload)
            {
#pragma unroll
                for (int r = 0; r < 4; r++)
                    acc[ni][mi][r] = FR_C(ni, mi, r);
            }
            else
                acc[ni][mi] = (v4f64){0.0, 0.0, 0.0, 0.0};
        }

    // prologue: STAGES - 1 slabs in flight
#pragma unroll
    for (int p = 0; p < STAGES - 1; p++)
        if (p < T)
            issue(p);

    // fragment addresses inside a stage (doubles): A (k, m) at k * FR_LDA + m;  B (k, n) at A_IMAGE + n * 16 + 2 (kp ^ s(n)) + (k & 1)
    const int a_frag = l4 * FR_LDA + wm + l15;
    int b_frag[2][2]; // [ni][pair of the quad: l4 >> 1 picks it]: position of (k = 4 kq + l4) for kq = 0 is computed, kq adds 2 to kp
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
    {
        const int n = wn + ni * 16 + l15;
        b_frag[ni][0] = FR_KS * FR_LDA + n * 16 + (l4 & 1);
        b_frag[ni][1] = (n >> 1) & 7; // swizzle of the column
    }

    for (int st = 0; st < T; st++)
    {
        // slab st has landed: this wave's own four instructions by the counted wait, everybody else's behind the barrier
        const int ahead = min(STAGES - 2, T - 1 - st); // slabs issued after slab st that may stay in flight
        if (STAGES >= 4 && ahead >= 2)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (STAGES >= 3 && ahead >= 1)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ... and every wave has finished reading the stage slab st - 1 was in: refill it
        if (st + STAGES - 1 < T)
            issue(st + STAGES - 1);
        const double *sA = lds + (st % STAGES) * FR_STAGE_DOUBLES;
#if PG_PLANES > 1
        const bool add = my_tasks[st >> steps_shift].sign < 0; // (complex updates as four real products: A_im B_im ADDS to the real plane)
#endif
#if PG_PLANES > 1
#define FR_MFMA(ni_, mi_, fb_, fa_)                                                                                        \
    if (add)                                                                                                                \
        acc[ni_][mi_] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb_, fa_, acc[ni_][mi_], 0, 0, 0);                          \
    else                                                                                                                    \
        acc[ni_][mi_] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb_, fa_, acc[ni_][mi_], 0, 0, DG_NEG_A);
#else
#define FR_MFMA(ni_, mi_, fb_, fa_) acc[ni_][mi_] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb_, fa_, acc[ni_][mi_], 0, 0, DG_NEG_A);
#endif
#define FR_READ(fa_, fb_, kq_)                                                                        \
    {                                                                                                 \
        _Pragma("unroll") for (int mi = 0; mi < 4; mi++)(fa_)[mi] = sA[a_frag + (kq_) * 4 * FR_LDA + mi * 16]; \
        _Pragma("unroll") for (int ni = 0; ni < 2; ni++)(fb_)[ni] = sA[b_frag[ni][0] + 2 * ((2 * (kq_) + (l4 >> 1)) ^ b_frag[ni][1])]; \
    }
        if (PREFETCH)
        {
            double fa[2][4], fb[2][2];
            FR_READ(fa[0], fb[0], 0)
#pragma unroll
            for (int kq = 0; kq < FR_KS / 4; kq++)
            {
                if (kq + 1 < FR_KS / 4)
                    FR_READ(fa[(kq + 1) & 1], fb[(kq + 1) & 1], kq + 1)
#pragma unroll
                for (int ni = 0; ni < 2; ni++)
#pragma unroll
                    for (int mi = 0; mi < 4; mi++)
                    {
                        FR_MFMA(ni, mi, fb[kq & 1][ni], fa[kq & 1][mi])
                    }
            }
        }
        else
        {
#pragma unroll
            for (int kq = 0; kq < FR_KS / 4; kq++)
            {
                double fa[4], fb[2];
                FR_READ(fa, fb, kq)
#pragma unroll
                for (int ni = 0; ni < 2; ni++)
#pragma unroll
                    for (int mi = 0; mi < 4; mi++)
                    {
                        FR_MFMA(ni, mi, fb[ni], fa[mi])
                    }
            }
        }
#undef FR_READ
#undef FR_MFMA
    }
    if (product_counter && lane == 0 && T)
        atomicAdd(product_counter, (unsigned long long)(8 * T)); // 16 x 16 x 16 products issued by this wave

    if (preload)
    {
        // the accumulators hold C - sum A B
#pragma unroll
        for (int ni = 0; ni < 2; ni++)
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    FR_C(ni, mi, r) = acc[ni][mi][r];
        return;
    }
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
    {
        if (G.atomic)
        {
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    atomicAdd((double *)&FR_C(ni, mi, r), acc[ni][mi][r]);
            continue;
        }
        double old[4][4];
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                old[mi][r] = FR_C(ni, mi, r);
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                FR_C(ni, mi, r) = old[mi][r] + acc[ni][mi][r];
    }
#undef FR_C
}

// ---------------------------------------------------------------------------------------------------------------
// The same pipeline for tiles with STRUCTURAL ZEROS: ssssm_tilesv_f64_kernel below.  (Its predecessor of round 3,
// ssssm_tiles_f64_kernel<STAGES>, lives in tools/experiments/ssssm_tiles.h since round 6.)  What the profile of the round-2 kernel
// said and what both kernels do about it: operand pieces that are structurally empty are not fetched; wavefront w owns the pieces
// (row piece 2 mi + (w & 1), column piece (w >> 1) + 4 ni) -- the same 4 + 2 fragment reads per k-quarter as a contiguous sub-tile,
// but a contiguous range of live rows or columns is spread over all wavefronts; per window of 16 queued updates one (task, K-slab)
// pair per thread is tested against the occupancy maps carried by the task descriptors and the live ones are compacted into a
// step list in LDS.
// ---------------------------------------------------------------------------------------------------------------
#define TL_WINDOW 16
#ifndef TL_MARK // (cycle probes of tools/microbench/front_gemm.hip; nothing in the product build)
#define TL_PROBE_DECL
#define TL_MARK(i)
#define TL_PROBE_STEP
#define TL_PROBE_FLUSH
#endif
#ifndef TL_ITEM
#define TL_ITEM_DECL
#define TL_ITEM(i)
#define TL_ITEM_COUNT
#endif
// ---------------------------------------------------------------------------------------------------------------
// ssssm_tilesv_f64_kernel: the two-stage general update kernel with the fixed cost of a slab step taken out of the chain.
// Cycle probes in the kernel above (tools/microbench/front_gemm.hip -DTL_PROBE, profiles/r03n_step_cost.log: steps with
// k x k live products of 64): a step costs F + W cycles, W = its matrix-core work and F = 2300 cycles that do not overlap
// with it -- the two workgroups of a CU fall into phase: 860 cycles between the barrier and the first fragment read (two
// dependent LDS round trips for the step word and the operand pointers, then all eight wavefronts queue their four DMA
// instructions at a 64 B/clk address path at once: 16 cycles each), a third round trip for the step word and a fourth for
// the first fragments before the first MFMA.
//  * step records (word, A pointer, B pointer) are fetched from LDS BEFORE the wait and the barrier of the previous step,
//    and arrive while the wavefront is parked there anyway;
//  * after the barrier: fragments of k-quarter 0, its products, THEN the DMA instructions for the next slab -- the address
//    path works in the shadow of the matrix cores and a wavefront's DMA instructions no longer queue behind fifteen others;
//  * DMA instructions whose B piece is dead are not issued (every wait of a two-stage pipeline is vmcnt(0)).
// ---------------------------------------------------------------------------------------------------------------
#define TV_STEPS (TL_WINDOW * 16)
__global__ __launch_bounds__(FR_THREADS, 4) void ssssm_tilesv_f64_kernel(const SsssmTaskD *__restrict__ tasks, int nb, const SsssmWorkD *__restrict__ work,
                                                                          unsigned long long *__restrict__ product_counter, unsigned unit)
{
    __shared__ __align__(16) double lds[2 * FR_STAGE_DOUBLES];
    __shared__ unsigned long long s_spa[TV_STEPS], s_spb[TV_STEPS]; // operand mirrors of the live steps of the window, in order
    __shared__ u32 s_step[TV_STEPS];                                // task << 20 | slab << 16 | bbits << 8 | abits
    __shared__ u32 s_cnt[FR_THREADS / 64];
#if PG_PLANES > 1
    __shared__ double s_sign[TL_WINDOW];
#endif
    const int tiles = nb / FR_TILE;
    const unsigned bid = logical_block_id(unit ? unit : (unsigned)(tiles * tiles));
    const SsssmWorkD G = work[bid];
    const int tile = (int)G.tile;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M0 = (tile % tiles) * FR_TILE, N0 = (tile / tiles) * FR_TILE;
    const int wr = wave & 1, wc = wave >> 1; // row pieces 2 mi + wr, column pieces wc + 4 ni
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ntask = (int)(G.task_end - G.task_begin);
    const int nslab = nb / FR_KS;
    const SsssmTaskD *my_tasks = tasks + G.task_begin;

    const unsigned a_voff = (unsigned)lane * 16u;
    const int a_piece = lane >> 3;
    const int bc = lane >> 3, bj = lane & 7;
    const unsigned b_voff = ((unsigned)bc * (unsigned)nb + 2u * (unsigned)(bj ^ ((4 * (wave & 1) + (bc >> 1)) & 7))) * 8u;

    v4f64 acc[2][4];
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
            acc[ni][mi] = (v4f64){0.0, 0.0, 0.0, 0.0};
    unsigned touched = 0, nprod = 0;

    const int a_frag = l4 * FR_LDA + wr * 16 + l15; // + kq * 4 * FR_LDA + mi * 32
    int b_frag[2][2];
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
    {
        const int n = (wc + 4 * ni) * 16 + l15;
        b_frag[ni][0] = FR_KS * FR_LDA + n * 16 + (l4 & 1);
        b_frag[ni][1] = (n >> 1) & 7;
    }
    const bool all_live = G.pad_ != 0;
    const int slab_shift = nb == 256 ? 4 : 3;

    auto uniform64 = [&](unsigned long long v) -> fr_gptr
    { return (fr_gptr)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v)); };
    auto issue = [&](unsigned w, fr_gptr pa, fr_gptr pb, int stage_no)
    {
        const unsigned ab = w & 0xFFu, bb = (w >> 8) & 0xFFu;
        const int k0 = (int)((w >> 16) & 15u) * FR_KS;
        double *stage = lds + stage_no * FR_STAGE_DOUBLES;
        const bool a_live = (ab >> a_piece) & 1u;
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
            const int k = wave + 8 * h;
            const unsigned off = a_live ? (unsigned)(((k0 + k) * nb + M0) * 8) + a_voff : 0u;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(dg_scalar_base(pa) + off), (fr_lptr)(stage + k * FR_LDA), 16, 0, 0);
        }
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
            const int g = wave + 8 * h;
            if (!((bb >> (g >> 1)) & 1u))
                continue;
            const unsigned off = (unsigned)(((N0 + 8 * g) * nb + k0) * 8) + b_voff;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(dg_scalar_base(pb) + off), (fr_lptr)(stage + FR_KS * FR_LDA + g * 128), 16, 0, 0);
        }
    };

    double __attribute__((address_space(1))) *C = (double __attribute__((address_space(1))) *)reinterpret_cast<double *>(G.cdense);
    const unsigned c_voff = ((unsigned)l4 * (unsigned)nb + (unsigned)l15) * 8u;
#define TL_C(ni_, mi_, r_)                                                                           \
    (*(double __attribute__((address_space(1))) *)(dg_scalar_base((const char __attribute__((address_space(1))) *)C +                    \
                                                                  ((size_t)(N0 + (wc + 4 * (ni_)) * 16 + 4 * (r_)) * nb + M0 + wr * 16) * 8) + \
                                                   dg_lane_offset(c_voff) + (mi_) * 256))
    unsigned pre = 0;
    const bool may_preload = !G.atomic && ntask <= TL_WINDOW;

    int stage_head = 0;
    TL_ITEM_DECL
    TL_ITEM_COUNT
    for (int win0 = 0; win0 < ntask; win0 += TL_WINDOW)
    {
        // ---- the window's step list (no DMA is in flight here: plain barriers) --------------------------------------
        __syncthreads();
        int T;
        if (all_live)
        {
            // step e = slab e & (nslab - 1) of task e >> slab_shift, every piece live
            T = min(TL_WINDOW, ntask - win0) * nslab;
            if (tid < T)
            {
                const int t_ = tid >> slab_shift;
                const SsssmTaskD &Tm = my_tasks[win0 + t_];
                s_step[tid] = ((unsigned)t_ << 20) | ((unsigned)(tid & (nslab - 1)) << 16) | 0xFFFFu;
                s_spa[tid] = (unsigned long long)reinterpret_cast<const double *>(Tm.a.val);
                s_spb[tid] = (unsigned long long)reinterpret_cast<const double *>(Tm.b.val);
#if PG_PLANES > 1
                if ((tid & (nslab - 1)) == 0)
                    s_sign[t_] = Tm.sign;
#endif
            }
            __syncthreads();
        }
        else
        {
            unsigned v = 0;
            unsigned long long pa_v = 0, pb_v = 0;
            const int t_ = tid >> 4, s_ = tid & 15;
            if (tid < TL_WINDOW * 16 && win0 + t_ < ntask && s_ < nslab)
            {
                const SsssmTaskD &Tm = my_tasks[win0 + t_];
                const double *pa_ = reinterpret_cast<const double *>(Tm.a.val), *pb_ = reinterpret_cast<const double *>(Tm.b.val);
                pa_v = (unsigned long long)pa_;
                pb_v = (unsigned long long)pb_;
                unsigned ab_, bb_ = 0;
                if (Tm.has_map)
                {
                    ab_ = ((unsigned)Tm.amap[s_] >> (M0 / 16)) & 0xFFu;
                    bb_ = ((unsigned)Tm.bmap_t[s_] >> (N0 / 16)) & 0xFFu;
                }
                else
                {
                    ab_ = ((unsigned)mirror_map(pa_, nb)[s_] >> (M0 / 16)) & 0xFFu;
                    const uint4 mb_ = *reinterpret_cast<const uint4 *>(mirror_map(pb_, nb) + N0 / 16);
                    const unsigned w_[4] = {mb_.x, mb_.y, mb_.z, mb_.w};
#pragma unroll
                    for (int c_ = 0; c_ < 8; c_++)
                        bb_ |= (((w_[c_ >> 1] >> (16 * (c_ & 1))) >> s_) & 1u) << c_;
                }
                if (ab_ && bb_ && (!G.slab_mask || ((G.slab_mask >> s_) & 1u)))
                    v = (bb_ << 8) | ab_ | ((unsigned)s_ << 16) | ((unsigned)t_ << 20);
#if PG_PLANES > 1
                if (s_ == 0)
                    s_sign[t_] = Tm.sign;
#endif
            }
            const unsigned long long bal = __ballot(v != 0);
            TL_ITEM(0)
            if (lane == 0)
                s_cnt[wave] = (u32)__builtin_popcountll(bal);
            __syncthreads();
            unsigned at = (unsigned)__builtin_popcountll(bal & ((1ull << lane) - 1ull)), all = 0;
#pragma unroll
            for (int w_i = 0; w_i < FR_THREADS / 64; w_i++)
            {
                const unsigned c_ = s_cnt[w_i];
                at += w_i < wave ? c_ : 0u;
                all += c_;
            }
            if (v)
            {
                s_step[at] = v;
                s_spa[at] = pa_v;
                s_spb[at] = pb_v;
            }
            __syncthreads();
            T = __builtin_amdgcn_readfirstlane((int)all);
            TL_ITEM(1)
        }
        if (T == 0)
            continue;

        // (the destination first: its loads are then older than every DMA, and the waits below cover them)
        if (may_preload)
        {
            unsigned m = 0;
            if (all_live)
                m = 0xFFu;
            else
            {
                for (int e = lane; e < T; e += 64)
                {
                    const unsigned w = s_step[e];
                    const unsigned ab = w & 0xFFu, bb = (w >> 8) & 0xFFu;
                    const unsigned a4 = ((ab >> wr) & 1u) | (((ab >> (2 + wr)) & 1u) << 1) | (((ab >> (4 + wr)) & 1u) << 2) | (((ab >> (6 + wr)) & 1u) << 3);
                    if ((bb >> wc) & 1u)
                        m |= a4;
                    if ((bb >> (wc + 4)) & 1u)
                        m |= a4 << 4;
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1)
                    m |= (unsigned)__shfl_xor((int)m, off, 64);
            }
            pre = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int mi = 0; mi < 4; mi++)
                    if ((pre >> (4 * ni + mi)) & 1u)
                    {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            acc[ni][mi][r] = TL_C(ni, mi, r);
                    }
        }
        // ---- the pipeline over the window's T live steps -------------------------------------------------------------
        // wC: word of the step being consumed; (wN, paN, pbN): record of the step whose slab is fetched during it
        TL_ITEM(2)
        unsigned wC = (unsigned)__builtin_amdgcn_readfirstlane((int)s_step[0]), wN = 0;
        fr_gptr paN = uniform64(s_spa[0]), pbN = uniform64(s_spb[0]);
        issue(wC, paN, pbN, stage_head);
        TL_ITEM(3)
        if (T > 1)
        {
            wN = (unsigned)__builtin_amdgcn_readfirstlane((int)s_step[1]);
            paN = uniform64(s_spa[1]);
            pbN = uniform64(s_spb[1]);
        }
        TL_PROBE_DECL
        for (int st = 0; st < T; st++)
        {
            TL_MARK(6)
            TL_PROBE_STEP
            // (record of step st + 2: in flight across the wait and the barrier)
            const int e2 = min(st + 2, T - 1);
            const unsigned v_w = s_step[e2];
            const unsigned long long v_pa = s_spa[e2], v_pb = s_spb[e2];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TL_MARK(0)
            __builtin_amdgcn_s_barrier();
            TL_MARK(1)
            const unsigned w2 = (unsigned)__builtin_amdgcn_readfirstlane((int)v_w);
            const fr_gptr pa2 = uniform64(v_pa), pb2 = uniform64(v_pb);
            const double *sA = lds + ((stage_head + st) & 1) * FR_STAGE_DOUBLES;
            const unsigned ab = wC & 0xFFu, bb = (wC >> 8) & 0xFFu;
            const unsigned a4 = ((ab >> wr) & 1u) | (((ab >> (2 + wr)) & 1u) << 1) | (((ab >> (4 + wr)) & 1u) << 2) | (((ab >> (6 + wr)) & 1u) << 3);
            const unsigned b2 = ((bb >> wc) & 1u) | (((bb >> (wc + 4)) & 1u) << 1);
            const bool live = a4 && b2;
#if PG_PLANES > 1
            const bool add = s_sign[wC >> 20] < 0;
#endif
#define TV_READ(buf_, kq_)                                                                            \
    {                                                                                                 \
        _Pragma("unroll") for (int mi = 0; mi < 4; mi++) fa[buf_][mi] = sA[a_frag + (kq_) * 4 * FR_LDA + mi * 32]; \
        _Pragma("unroll") for (int ni = 0; ni < 2; ni++) fb[buf_][ni] = sA[b_frag[ni][0] + 2 * ((2 * (kq_) + (l4 >> 1)) ^ b_frag[ni][1])]; \
    }
            double fa[2][4], fb[2][2];
            // Round 6: PIECE-major inside a pair of k-quarters: with both k-quarters of a pair in registers a live piece takes its two
            // products behind ONE test -- 16 tests per step instead of 40; a wavefront's step went from 122 scalar instructions and
            // 79 branches to 79 and 41 (ISA count).  Measured: + 1 % stand-alone (tools/microbench/front_gemm.hip, profiles/r06g*),
            // nothing in the factorisation (1526.0 against 1526.2 ms, profiles/r06h_*): the step loop is NOT bound by scalar issue.
            // Kept because it is the shorter program.
#define TV_PIECE2(ni_, mi_)                                                                                                            \
    if ((m8 >> ((mi_) + 4 * (ni_))) & 1u)                                                                                              \
    {                                                                                                                                   \
        TV_ONE(0, ni_, mi_)                                                                                                             \
        TV_ONE(1, ni_, mi_)                                                                                                             \
    }
#if PG_PLANES > 1
#define TV_ONE(buf_, ni_, mi_)                                                                                                          \
    if (add)                                                                                                                            \
        acc[ni_][mi_] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[buf_][ni_], fa[buf_][mi_], acc[ni_][mi_], 0, 0, 0);                     \
    else                                                                                                                                \
        acc[ni_][mi_] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[buf_][ni_], fa[buf_][mi_], acc[ni_][mi_], 0, 0, DG_NEG_A);
#else
#define TV_ONE(buf_, ni_, mi_) acc[ni_][mi_] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[buf_][ni_], fa[buf_][mi_], acc[ni_][mi_], 0, 0, DG_NEG_A);
#endif
#define TV_PAIR                                                                            \
    {                                                                                      \
        TV_PIECE2(0, 0) TV_PIECE2(0, 1) TV_PIECE2(0, 2) TV_PIECE2(0, 3)                    \
        TV_PIECE2(1, 0) TV_PIECE2(1, 1) TV_PIECE2(1, 2) TV_PIECE2(1, 3)                    \
    }
            unsigned m8 = 0; // bit mi + 4 ni: this wavefront's live pieces in the step
            if (live)
            {
                m8 = ((b2 & 1u) ? a4 : 0u) | ((b2 & 2u) ? (a4 << 4) : 0u);
                nprod += (unsigned)__builtin_popcount(m8);
                touched |= m8;
                TV_READ(0, 0)
                TV_READ(1, 1)
                TV_PAIR
            }
            TL_MARK(2)
            // the next slab, behind this wavefront's first products
            if (st + 1 < T)
                issue(wN, paN, pbN, (stage_head + st + 1) & 1);
            TL_MARK(3)
            if (live)
            {
                TV_READ(0, 2)
                TV_READ(1, 3)
                TV_PAIR
            }
#undef TV_PIECE2
#undef TV_ONE
#undef TV_PAIR
#undef TV_READ
            wC = wN;
            wN = w2;
            paN = pa2;
            pbN = pb2;
            TL_MARK(4)
        }
        TL_PROBE_FLUSH
        stage_head = (stage_head + T) & 1;
        TL_ITEM(4)
    }
    if (product_counter && lane == 0 && nprod)
        atomicAdd(product_counter, (unsigned long long)nprod);

    if (pre)
    {
#pragma unroll
        for (int ni = 0; ni < 2; ni++)
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
                if ((pre >> (4 * ni + mi)) & 1u)
                {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        TL_C(ni, mi, r) = acc[ni][mi][r];
                }
        touched = 0;
    }
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
    {
        const unsigned t4 = (touched >> (4 * ni)) & 0xFu;
        if (!t4)
            continue;
        if (G.atomic)
        {
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
            {
                if (!((t4 >> mi) & 1u))
                    continue;
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (acc[ni][mi][r] != 0.0)
                        atomicAdd((double *)&TL_C(ni, mi, r), acc[ni][mi][r]);
            }
            continue;
        }
        double old[4][4];
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                old[mi][r] = ((t4 >> mi) & 1u) ? TL_C(ni, mi, r) : 0.0;
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
        {
            if (!((t4 >> mi) & 1u))
                continue;
#pragma unroll
            for (int r = 0; r < 4; r++)
                TL_C(ni, mi, r) = old[mi][r] + acc[ni][mi][r];
        }
    }
#ifdef TL_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TL_ITEM(5)
#endif
#undef TL_C
}
