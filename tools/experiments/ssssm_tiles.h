// tools/experiments/ssssm_tiles.h -- ssssm_tiles_f64_kernel<STAGES>: round 3's first LDS-DMA general update kernel (DMA issue right
// behind the barrier, step words through LDS, STAGES = 2 / 3 / 4).  Superseded by ssssm_tilesv_f64_kernel (pg_hip_front.h: step
// records a step ahead, DMA issue behind the first products), which won every comparison since (profiles/r03n_step_cost.log,
// r03*_tiles_*); moved out of the product in round 6.  tools/microbench/front_gemm.hip still checks and times
// it: include AFTER pg_hip_front.h (it uses its constants, fr_gptr / fr_lptr and the TL_* probe macros).
#pragma once
// ---------------------------------------------------------------------------------------------------------------
// The same pipeline for tiles with STRUCTURAL ZEROS (round 3: the general MFMA update kernel).
//
// What the profile of the round-2 kernel said (fem27(112), PANGULU_HIP_LAUNCH_LOG): its launches process LIVE SLAB STEPS
// (a 128 x 128 x 16 product with at least one live piece on either side) at the rate a full slab would take at 93 % of the
// f64 MFMA peak -- whatever is inside: only 55 % of the 16 x 16 x 16 products of those steps are live.  A step costs one
// memory round trip (one slab in flight per workgroup, two workgroups per CU) plus a barrier at which everybody waits for
// the wavefront with the most products: with contiguous 64 x 32 sub-tiles per wavefront that one has 5.8 of 8 on average
// where the mean is 3.5 (fill patterns are made of contiguous ranges of rows and columns).
//  * Pipeline: LDS-DMA, STAGES deep, as above.  Pieces that are structurally empty are not fetched: their lanes (A: eight
//    lanes per 16-row piece of a column) or their whole instruction (B: half a 16-column piece) read 16 bytes from the
//    start of the mirror instead -- an L2 hit that lands in a part of the image no MFMA reads -- so that every wave still
//    issues exactly four DMA instructions per step and the waits stay counted.
//  * Ownership: wavefront w owns the pieces (row piece 2 mi + (w & 1), column piece (w >> 1) + 4 ni), mi < 4, ni < 2 -- the
//    same 4 + 2 fragment reads per k-quarter as a contiguous sub-tile, but a contiguous range of live rows or columns is
//    spread over all wavefronts (simulated on fem27(40)'s patterns: busiest wavefront 4.7 products per step instead of 5.8).
//  * Bookkeeping as in round 2: per window of 16 queued updates one (task, K-slab) pair per thread is tested against the
//    occupancy maps carried by the task descriptors and the live ones are compacted into a step list in LDS; the pipeline
//    drains between windows (queues longer than 16 are rare: the scheduler flushes them level by level).
// ---------------------------------------------------------------------------------------------------------------
template <int STAGES>
__global__ __launch_bounds__(FR_THREADS, (STAGES <= 2 ? 4 : 2)) void ssssm_tiles_f64_kernel(const SsssmTaskD *__restrict__ tasks, int nb, const SsssmWorkD *__restrict__ work,
                                                                     unsigned long long *__restrict__ product_counter, unsigned unit)
{
    __shared__ __align__(16) double lds[STAGES * FR_STAGE_DOUBLES];
    __shared__ u32 s_step[TL_WINDOW * 16]; // live steps of the window in order: task << 20 | slab << 16 | bbits << 8 | abits
    __shared__ u32 s_cnt[FR_THREADS / 64];
    __shared__ unsigned long long s_pa[TL_WINDOW], s_pb[TL_WINDOW];
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
    const int a_piece = lane >> 3; // the 16-row piece this lane's 16 bytes of an A column belong to
    const int bc = lane >> 3, bj = lane & 7;
    unsigned b_voff[2];
#pragma unroll
    for (int par = 0; par < 2; par++)
        b_voff[par] = ((unsigned)bc * (unsigned)nb + 2u * (unsigned)(bj ^ ((4 * par + (bc >> 1)) & 7))) * 8u;

    v4f64 acc[2][4];
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
            acc[ni][mi] = (v4f64){0.0, 0.0, 0.0, 0.0};
    unsigned touched = 0, nprod = 0;

    // fragment addresses inside a stage (doubles)
    const int a_frag = l4 * FR_LDA + wr * 16 + l15; // + kq * 4 * FR_LDA + mi * 32
    int b_frag[2][2];
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
    {
        const int n = (wc + 4 * ni) * 16 + l15;
        b_frag[ni][0] = FR_KS * FR_LDA + n * 16 + (l4 & 1);
        b_frag[ni][1] = (n >> 1) & 7;
    }

    // A work item whose whole queue is dense-front products (the host sets pad_ = 1: every 16 x 16 piece of every operand that
    // meets the tile is live, no K-split) needs no step list: step e of a window is slab e & 15 of task e >> 4, all pieces live.
    // Such items share the launch with the partly filled ones (one launch = one tail) and skip the bookkeeping barriers.
    const bool all_live = G.pad_ != 0;
    const int slab_shift = nb == 256 ? 4 : 3;
    // step e of the current window as scalars
    auto step_word = [&](int e) -> unsigned
    {
        if (all_live)
            return ((unsigned)(e >> slab_shift) << 20) | ((unsigned)(e & (nslab - 1)) << 16) | 0xFFFFu; // (nslab = 8 or 16)
        return (unsigned)__builtin_amdgcn_readfirstlane((int)s_step[e]);
    };
    auto task_ptr = [&](const unsigned long long *tab, unsigned t) -> fr_gptr
    {
        const unsigned long long v = tab[t];
        return (fr_gptr)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
    };
    auto issue = [&](int e, int stage_no)
    {
        const unsigned w = step_word(e);
        const unsigned ab = w & 0xFFu, bb = (w >> 8) & 0xFFu, t = w >> 20;
        const int k0 = (int)((w >> 16) & 15u) * FR_KS;
        const fr_gptr pa = task_ptr(s_pa, t), pb = task_ptr(s_pb, t);
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
            const bool b_live = (bb >> (g >> 1)) & 1u;
            const unsigned off = b_live ? (unsigned)(((N0 + 8 * g) * nb + k0) * 8) + b_voff[wave & 1] : 0u;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(dg_scalar_base(pb) + off), (fr_lptr)(stage + FR_KS * FR_LDA + g * 128), 16, 0, 0);
        }
    };

    // register r of lane l of piece (ni, mi) is C(M0 + (2 mi + wr) 16 + (l & 15), N0 + (wc + 4 ni) 16 + 4 r + (l >> 4))
    double __attribute__((address_space(1))) *C = (double __attribute__((address_space(1))) *)reinterpret_cast<double *>(G.cdense);
    const unsigned c_voff = ((unsigned)l4 * (unsigned)nb + (unsigned)l15) * 8u;
#define TL_C(ni_, mi_, r_)                                                                           \
    (*(double __attribute__((address_space(1))) *)(dg_scalar_base((const char __attribute__((address_space(1))) *)C +                    \
                                                                  ((size_t)(N0 + (wc + 4 * (ni_)) * 16 + 4 * (r_)) * nb + M0 + wr * 16) * 8) + \
                                                   dg_lane_offset(c_voff) + (mi_) * 256))
    // A workgroup that owns its destination (no atomics) and whose queue fits one window knows, once the step list is there,
    // which pieces of C it is going to touch: their values go into the accumulators before the first slab is consumed (the
    // matrix cores subtract: acc = C - sum A B) and the epilogue is stores only -- a read-modify-write at the end is a dependent
    // memory round trip that nothing hides.
    unsigned pre = 0;
    const bool may_preload = !G.atomic && ntask <= TL_WINDOW;

    int stage_head = 0; // stage the next consumed step sits in (stages are used round-robin across windows)
    for (int win0 = 0; win0 < ntask; win0 += TL_WINDOW)
    {
        // ---- the window's step list (no DMA is in flight here: plain barriers) --------------------------------------
        __syncthreads();
        unsigned v = 0;
        if (all_live)
        {
            // (only the operand pointers of the window's tasks)
            if (tid < TL_WINDOW && win0 + tid < ntask)
            {
                const SsssmTaskD &Tm = my_tasks[win0 + tid];
                s_pa[tid] = (unsigned long long)reinterpret_cast<const double *>(Tm.a.val);
                s_pb[tid] = (unsigned long long)reinterpret_cast<const double *>(Tm.b.val);
#if PG_PLANES > 1
                s_sign[tid] = Tm.sign;
#endif
            }
        }
        else
        {
            const int t_ = tid >> 4, s_ = tid & 15;
            if (tid < TL_WINDOW * 16 && win0 + t_ < ntask && s_ < nslab)
            {
                const SsssmTaskD &Tm = my_tasks[win0 + t_];
                const double *pa_ = reinterpret_cast<const double *>(Tm.a.val), *pb_ = reinterpret_cast<const double *>(Tm.b.val);
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
                if (s_ == 0)
                {
                    s_pa[t_] = (unsigned long long)pa_;
                    s_pb[t_] = (unsigned long long)pb_;
#if PG_PLANES > 1
                    s_sign[t_] = Tm.sign;
#endif
                }
            }
        }
        int T;
        if (all_live)
        {
            __syncthreads();
            T = min(TL_WINDOW, ntask - win0) * nslab;
        }
        else
        {
            const unsigned long long bal = __ballot(v != 0);
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
                s_step[at] = v;
            __syncthreads();
            T = __builtin_amdgcn_readfirstlane((int)all);
        }
        if (T == 0)
            continue;

        // (the destination first: its loads are then older than every DMA, and the counted waits below cover them)
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
#pragma unroll
        for (int p = 0; p < STAGES - 1; p++)
            if (p < T)
                issue(p, (stage_head + p) % STAGES);
        TL_PROBE_DECL
        for (int st = 0; st < T; st++)
        {
            TL_MARK(4)
            TL_PROBE_STEP
            const int ahead = min(STAGES - 2, T - 1 - st);
            if (STAGES >= 4 && ahead >= 2)
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (STAGES >= 3 && ahead >= 1)
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TL_MARK(0)
            __builtin_amdgcn_s_barrier();
            TL_MARK(1)
            if (st + STAGES - 1 < T)
                issue(st + STAGES - 1, (stage_head + st + STAGES - 1) % STAGES);
            TL_MARK(2)
            const double *sA = lds + ((stage_head + st) % STAGES) * FR_STAGE_DOUBLES;
            const unsigned w = step_word(st);
            const unsigned ab = w & 0xFFu, bb = (w >> 8) & 0xFFu;
            // this wavefront's live pieces: rows 2 mi + wr, columns wc + 4 ni
            const unsigned a4 = ((ab >> wr) & 1u) | (((ab >> (2 + wr)) & 1u) << 1) | (((ab >> (4 + wr)) & 1u) << 2) | (((ab >> (6 + wr)) & 1u) << 3);
            const unsigned b2 = ((bb >> wc) & 1u) | (((bb >> (wc + 4)) & 1u) << 1);
            if (a4 && b2)
            {
                nprod += (unsigned)(__builtin_popcount(a4) * __builtin_popcount(b2));
#pragma unroll
                for (int ni = 0; ni < 2; ni++)
                    if ((b2 >> ni) & 1u)
                        touched |= a4 << (4 * ni);
#if PG_PLANES > 1
                const bool add = s_sign[w >> 20] < 0;
#endif
                // fragments of k-quarter kq + 1 are read while the matrix cores work on kq (dead pieces are read too: the
                // addresses are valid LDS, the values are never used)
                double fa[2][4], fb[2][2];
#define TL_READ(buf_, kq_)                                                                            \
    {                                                                                                 \
        _Pragma("unroll") for (int mi = 0; mi < 4; mi++) fa[buf_][mi] = sA[a_frag + (kq_) * 4 * FR_LDA + mi * 32]; \
        _Pragma("unroll") for (int ni = 0; ni < 2; ni++) fb[buf_][ni] = sA[b_frag[ni][0] + 2 * ((2 * (kq_) + (l4 >> 1)) ^ b_frag[ni][1])]; \
    }
                TL_READ(0, 0)
#pragma unroll
                for (int kq = 0; kq < FR_KS / 4; kq++)
                {
                    if (kq + 1 < FR_KS / 4)
                        TL_READ((kq + 1) & 1, kq + 1)
#pragma unroll
                    for (int ni = 0; ni < 2; ni++)
                    {
                        if (!((b2 >> ni) & 1u))
                            continue;
#pragma unroll
                        for (int mi = 0; mi < 4; mi++)
                            if ((a4 >> mi) & 1u)
                            {
#if PG_PLANES > 1
                                if (add)
                                    acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[kq & 1][ni], fa[kq & 1][mi], acc[ni][mi], 0, 0, 0);
                                else
#endif
                                    acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[kq & 1][ni], fa[kq & 1][mi], acc[ni][mi], 0, 0, DG_NEG_A);
                            }
                    }
                }
#undef TL_READ
            }
            TL_MARK(3)
        }
        TL_PROBE_FLUSH
        stage_head = (stage_head + T) % STAGES;
    }
    if (product_counter && lane == 0 && nprod)
        atomicAdd(product_counter, (unsigned long long)nprod);

    if (pre)
    {
        // preloaded: the accumulators hold C - sum A B; stores only
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
        touched = 0; // (a subset of pre: everything has been written)
    }
    // C += acc on the touched pieces
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
#undef TL_C
}

