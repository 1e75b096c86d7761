// pg_hip_pieces.h -- ssssm_tilesp_f64_kernel: the general MFMA update kernel with COMPACTED, piece-indexed LDS staging
// (round 5; included by pg_hip_platform.hip after pg_hip_front.h; tools/microbench/front_gemm.hip times it stand-alone).
//
// Replaces ...0201000.cu:717-873 (per-task `ssssm_cuda` / cuBLAS on densified blocks); semantics of the CPU operator
// ...0100000.c:211-397: C -= A B restricted to C's pattern, here on the dense mirrors with the operands' 16 x 16 occupancy maps.
//
// What the tilesv kernel (pg_hip_front.h) left on the table (profiles/r04ao_elastic3d_77.md: 654 ms of 1 588 at 56.7 % of the
// matrix pipes): its staging is POSITIONAL -- piece r of a slab lands on rows 16 r.. of a 34 KB stage whatever the slab holds --
// so a slab step with 3 x 4 live pieces owns a whole stage and a whole barrier interval: a step costs F + W cycles with
// F = 2000-2300 (the loaded memory round trip of its DMA plus the barrier chain) whatever W, its matrix-core work, is
// (profiles/r03n_step_cost.log), and with the histogram of live products per step flat from 1 to 64, F is half the time.
//
// This kernel stages PIECES.  The unit of LDS is a 2 KiB slot holding one live 16 x 16 piece of an operand; the slots form a
// ring of 32 (64 KB, as much as the two positional stages), filled in step order: the live pieces of a step take consecutive
// slots (A pieces by row, then B pieces by column).  Consecutive live steps form a BATCH while their pieces fill at most 16 slots
// (a completely live 8 + 8 step is a batch of its own -- that case runs as before); a batch is what one barrier interval consumes:
//   * light steps share a barrier: 4 steps of 2 x 2 pieces are one batch, one wait, one barrier, 16 products between them;
//   * the ring holds the batch being consumed plus whatever of the following ones fits, so light batches run further ahead
//     where the positional stages had one slab in flight regardless;
//   * only live pieces are fetched (dead A pieces were fetched as 16-byte dummies, whole DMA instructions regardless).
// No tables beyond the tilesv kernel's list of live step words: every wavefront keeps the pipeline state in SCALAR registers and
// derives everything from the step words with scalar instructions -- which batch ends where (decided when its steps are issued and
// queued in a four-entry scalar FIFO for the consumer side of the same wavefront), which slot a piece has (the running count of
// pieces + a population count of the mask bits below it), how many DMA instructions of a batch are this wavefront's (wavefront w
// fetches the pieces whose ring position is w modulo 8: it counts as it issues).  Every vector instruction beside the f64 matrix pipe is
// paid in matrix-pipe time (pg_hip_dense.h) -- also those of a co-resident workgroup's bookkeeping: measured on this kernel's
// second version, whose per-step tables cost 24 000 cycles per work item to build beside a workgroup in its product loop.
//
// Piece images (decided by the per-lane SOURCE address of the DMA, which writes 64 x 16 bytes contiguously):
//   A piece (16 rows m x 16 columns k, mirror column-major): natural, column k at byte 128 k; fragment of k-quarter kq for lane
//     (m = l & 15, k = 4 kq + (l >> 4)) at 512 kq + 128 (l >> 4) + 8 m: conflict-free, kq an immediate;
//   B piece (16 rows k x 16 columns n): instruction g takes k = 8 g .. 8 g + 7, column n at byte 1024 g + 64 n, its four 16-byte
//     k-pairs XOR-swizzled by bit 2 of n on the source side (a quad of lanes still reads one 64-byte run); fragment at
//     1024 (kq >> 1) + 32 (kq & 1) + 64 n + 16 ((l >> 5) ^ ((n >> 2) & 1)) + 8 ((l >> 4) & 1): kq an immediate again, two lanes per
//     bank pair (a 2-way conflict on a quarter of the reads; a conflict-free swizzle would need kq inside the XOR, i.e. a vector
//     add per fragment read).
// So a live piece costs its wavefronts ONE vector add (slot base + lane constant) per step, whatever kq.
#pragma once

#define TP_RING 32             // ring slots (power of two)
#define TP_SLOT_BYTES 2048     // one 16 x 16 piece of doubles
#define TP_BATCH_SLOTS 16      // most slots a batch may take (>= 16: a completely live step)
#define TP_FIFO 4              // closed batches a wavefront may hold between its issue side and its consumer side
#ifndef TP_BATCH_PRODUCTS
#define TP_BATCH_PRODUCTS 40   // a batch is closed once it holds this many 16 x 16 x 16 products (enough work to cover a round trip)
#endif

typedef const double __attribute__((address_space(3))) *tp_lds_cd;

// s_waitcnt vmcnt(n) for a wave-uniform n known at run time (never waits for less than asked: n is rounded DOWN to a case)
__device__ __forceinline__ void tp_wait_vmcnt(unsigned n)
{
    if (n >= 12)
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n >= 8)
    {
        if (n >= 10)
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    else if (n >= 4)
    {
        if (n >= 6)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    else if (n >= 2)
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ unsigned tp_uniform(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
// OR over the 64 lanes of a wavefront, result in lane 63: six DPP steps (quad permutes, row mirrors, row broadcasts).  (Measured on this
// kernel: an LDS atomic from every lane to one address cost 7 000 - 19 000 cycles per work item.)
__device__ __forceinline__ unsigned tp_wave_or(unsigned v)
{
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true); // row_half_mirror
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true); // row_mirror
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true); // row_bcast:15 into rows 1 and 3
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true); // row_bcast:31 into rows 2 and 3
    return v;
}
// position of the r-th set bit of a 16-bit mask (r < popcount(m)), scalar: two halving steps by population counts, then at most
// three clear-lowest-bit steps
__device__ __forceinline__ unsigned tp_select_bit(unsigned m, unsigned r)
{
    unsigned base = 0;
    unsigned c = (unsigned)__builtin_popcount(m & 0xFFu);
    if (r >= c)
    {
        r -= c;
        m >>= 8;
        base = 8;
    }
    m &= 0xFFu;
    c = (unsigned)__builtin_popcount(m & 0xFu);
    if (r >= c)
    {
        r -= c;
        m >>= 4;
        base += 4;
    }
    m &= 0xFu;
    for (; r; r--)
        m &= m - 1u;
    return base + (unsigned)__builtin_ctz(m);
}
// (a value the compiler must keep in a scalar register: without the pin it re-derives uniform values in vector registers)
__device__ __forceinline__ unsigned tp_scalar(unsigned v)
{
    asm("" : "+s"(v));
    return v;
}

__global__ __launch_bounds__(FR_THREADS, 4) void ssssm_tilesp_f64_kernel(const SsssmTaskD *__restrict__ tasks, int nb, const SsssmWorkD *__restrict__ work,
                                                                          unsigned long long *__restrict__ product_counter, unsigned unit)
{
    __shared__ __align__(2048) double ring[TP_RING * (TP_SLOT_BYTES / 8)];
    __shared__ u32 s_word[TV_STEPS];                          // live steps of the window in order: task << 20 | slab << 16 | bbits << 8 | abits
    __shared__ unsigned long long s_pa[TL_WINDOW], s_pb[TL_WINDOW]; // operand mirrors of the window's tasks
    __shared__ u32 s_cnt[FR_THREADS / 64];
    __shared__ u32 s_touch[2]; // C pieces the queue touches: bit i + 8 j of the pair = piece (row i, column j)
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
    const unsigned ring_base = (unsigned)(unsigned long long)(fr_lptr)ring;

    // per-lane source offsets of a piece's first DMA instruction (the second one adds 8 columns (A) / 64 bytes (B) on the scalar side)
    const unsigned a_voff = (unsigned)(lane >> 3) * (unsigned)nb * 8u + (unsigned)(lane & 7) * 16u;
    const unsigned b_voff = (unsigned)(lane >> 2) * (unsigned)nb * 8u + (unsigned)((lane & 3) ^ ((lane >> 4) & 1)) * 16u;
    // per-lane fragment offsets inside a slot (+ the ring's base)
    const unsigned a_frag = ring_base + (unsigned)l4 * 128u + (unsigned)l15 * 8u;
    const unsigned b_frag = ring_base + (unsigned)l15 * 64u + (unsigned)((l4 >> 1) ^ ((l15 >> 2) & 1)) * 16u + (unsigned)(l4 & 1) * 8u;

    v4f64 acc[2][4];
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
            acc[ni][mi] = (v4f64){0.0, 0.0, 0.0, 0.0};
    unsigned touched = 0, nprod = 0;

    const bool all_live = G.pad_ != 0;

    double __attribute__((address_space(1))) *C = (double __attribute__((address_space(1))) *)reinterpret_cast<double *>(G.cdense);
    const unsigned c_voff = ((unsigned)l4 * (unsigned)nb + (unsigned)l15) * 8u;
#define TP_C(ni_, mi_, r_)                                                                           \
    (*(double __attribute__((address_space(1))) *)(dg_scalar_base((const char __attribute__((address_space(1))) *)C +                    \
                                                                  ((size_t)(N0 + (wc + 4 * (ni_)) * 16 + 4 * (r_)) * nb + M0 + wr * 16) * 8) + \
                                                   dg_lane_offset(c_voff) + (mi_) * 256))
    unsigned pre = 0;
    const bool may_preload = !G.atomic && ntask <= TL_WINDOW;

    TL_ITEM_DECL
    TL_ITEM_COUNT
    for (int win0 = 0; win0 < ntask; win0 += TL_WINDOW)
    {
        // ---- the window's list of live steps (no DMA is in flight here: plain barriers) ---------------------------------
        __syncthreads();
        if (tid < 2)
            s_touch[tid] = 0;
        unsigned v = 0, ab_ = 0, bb_ = 0;
        const int t_ = tid >> 4, s_ = tid & 15;
        if (tid < TL_WINDOW * 16 && win0 + t_ < ntask && s_ < nslab)
        {
            const SsssmTaskD &Tm = my_tasks[win0 + t_];
            const double *pa_ = reinterpret_cast<const double *>(Tm.a.val), *pb_ = reinterpret_cast<const double *>(Tm.b.val);
            if (all_live)
                ab_ = bb_ = 0xFFu;
            else if (Tm.has_map)
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
        const unsigned long long bal = __ballot(v != 0);
        TL_ITEM(0)
        if (lane == 0)
            s_cnt[wave] = (u32)__builtin_popcountll(bal);
        __syncthreads();
        unsigned at = (unsigned)__builtin_popcountll(bal & ((1ull << lane) - 1ull)), all = 0;
#pragma unroll
        for (int w_i = 0; w_i < 4; w_i++) // (candidates sit in the first four wavefronts)
        {
            const unsigned c_ = s_cnt[w_i];
            at += w_i < wave ? c_ : 0u;
            all += c_;
        }
        if (v)
            s_word[at] = v;
        // the pieces of C this queue touches (a work item that owns its destination takes them into the accumulators up front): the
        // outer product of the two piece masks (bytes of bb spread by a multiplication), OR-ed over the wavefront by DPP
        if (may_preload && wave < 4)
        {
            unsigned lo = v ? ab_ * (((bb_ & 0xFu) * 0x00204081u) & 0x01010101u) : 0u, hi = v ? ab_ * (((bb_ >> 4) * 0x00204081u) & 0x01010101u) : 0u;
            lo = tp_wave_or(lo);
            hi = tp_wave_or(hi);
            if (lane == 63)
            {
                atomicOr(&s_touch[0], lo);
                atomicOr(&s_touch[1], hi);
            }
        }
        __syncthreads();
        const int T = (int)tp_uniform(all);
        TL_ITEM(1)
        if (T == 0)
            continue;

        // (the destination first: its loads are then older than every DMA, and the counted waits below cover them)
        if (may_preload)
        {
            const unsigned lo = tp_uniform(s_touch[0]), hi = tp_uniform(s_touch[1]);
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int mi = 0; mi < 4; mi++)
                {
                    const int i = 2 * mi + wr, j = wc + 4 * ni;
                    const unsigned word = ni ? hi : lo; // (j >= 4 <=> ni = 1)
                    pre |= ((word >> (8 * (j & 3) + i)) & 1u) << (4 * ni + mi);
                }
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int mi = 0; mi < 4; mi++)
                    if ((pre >> (4 * ni + mi)) & 1u)
                    {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            acc[ni][mi][r] = TP_C(ni, mi, r);
                    }
        }
        TL_ITEM(2)

        // ---- the pipeline over the window's live steps -----------------------------------------------------------------
        // Issue side: step ib is the next to fetch, Pi pieces have been given slots; the OPEN batch (the one steps are being added
        // to) holds open_slots pieces, open_prods products and open_cnt DMA instructions of this wavefront; a batch is closed --
        // pushed into the FIFO as (end step | this wavefront's DMA instructions << 16) -- when the next step would overflow it.
        // Consumer side: batch = steps cb .. ce - 1, Pc pieces consumed before it.  S_issue / S_wait: DMA instructions issued /
        // waited for by this wavefront, in order.
        int ib = 0, cb = 0;
        unsigned Pi = 0, Pc = 0, S_issue = 0, S_wait = 0;
        unsigned open_slots = 0, open_prods = 0, open_cnt = 0;
        unsigned fifo[TP_FIFO] = {0, 0, 0, 0};
        int nfifo = 0;
        int cur_task = -1;
        fr_gptr pa_cur = nullptr, pb_cur = nullptr;
        unsigned wi = tp_uniform(s_word[0]);           // word of step ib
        unsigned wn_v = s_word[min(1, T - 1)];         // word of step ib + 1, in flight
        auto issue_piece = [&](fr_gptr src, bool is_b, unsigned slot)
        {
            const unsigned slot_addr = ring_base + slot * TP_SLOT_BYTES;
            const unsigned voff = is_b ? b_voff : a_voff;
            const unsigned long long half = is_b ? 64ull : (unsigned long long)nb * 64ull; // (A: eight columns further)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(dg_scalar_base(src) + voff), (fr_lptr)(unsigned long long)slot_addr, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(dg_scalar_base(src + half) + voff),
                                             (fr_lptr)(unsigned long long)(slot_addr + 1024u), 16, 0, 0);
        };
        auto try_issue = [&]()
        {
            while (ib < T && nfifo < TP_FIFO)
            {
                const unsigned ab = wi & 0xFFu, bb = (wi >> 8) & 0xFFu;
                const unsigned na = (unsigned)__builtin_popcount(ab), nbp = (unsigned)__builtin_popcount(bb), need = na + nbp;
                if (Pi + need - Pc > TP_RING)
                    break; // (the ring is full: the slots of the batch being consumed are free behind the next barrier)
                const int task = (int)(wi >> 20);
                if (task != cur_task)
                {
                    const unsigned long long va_ = s_pa[task], vb_ = s_pb[task];
                    pa_cur = (fr_gptr)(((unsigned long long)tp_uniform((unsigned)(va_ >> 32)) << 32) | tp_uniform((unsigned)va_));
                    pb_cur = (fr_gptr)(((unsigned long long)tp_uniform((unsigned)(vb_ >> 32)) << 32) | tp_uniform((unsigned)vb_));
                    cur_task = task;
                }
                const unsigned k0 = ((wi >> 16) & 15u) * FR_KS;
                // this wavefront's pieces of the step: the ring positions = its number modulo 8 (the r-th live piece of the step sits
                // at position Pi + r: A pieces by row, then B pieces by column)
                for (unsigned r = ((unsigned)wave - Pi) & 7u; r < need; r += 8u)
                {
                    const unsigned bit = tp_select_bit(ab | (bb << 8), r);
                    if (bit < 8u)
                        issue_piece(pa_cur + ((size_t)k0 * nb + M0 + 16 * bit) * 8, false, (Pi + r) & (TP_RING - 1));
                    else
                        issue_piece(pb_cur + ((size_t)(N0 + 16 * (bit - 8u)) * nb + k0) * 8, true, (Pi + r) & (TP_RING - 1));
                    open_cnt += 2;
                    S_issue += 2;
                }
                Pi += need;
                open_slots += need;
                open_prods += na * nbp;
                ib++;
                // the next step's word (requested an iteration ago); close the batch when that step would overflow it
                wi = tp_uniform(wn_v);
                wn_v = s_word[min(ib + 1, T - 1)];
                const unsigned need_next = (unsigned)(__builtin_popcount(wi & 0xFFu) + __builtin_popcount((wi >> 8) & 0xFFu));
                if (ib == T || open_slots + need_next > TP_BATCH_SLOTS || open_prods >= TP_BATCH_PRODUCTS)
                {
                    fifo[nfifo++] = (unsigned)ib | (open_cnt << 16);
                    open_slots = open_prods = open_cnt = 0;
                }
            }
        };
        try_issue();
        TL_ITEM(3)
        unsigned wc_v = s_word[0]; // word of the consumer's next step, in flight
        TL_PROBE_DECL
        while (cb < T)
        {
            TL_MARK(6)
            // the oldest closed batch (there is one: the batch behind the one being consumed is always completed by the issue side
            // in the middle of that one -- it fits the ring beside it by construction)
            const unsigned f0 = fifo[0];
#pragma unroll
            for (int q = 0; q + 1 < TP_FIFO; q++)
                fifo[q] = fifo[q + 1];
            nfifo--;
            const int ce = (int)(f0 & 0xFFFFu);
            S_wait += f0 >> 16;
            tp_wait_vmcnt(S_issue - S_wait);
            TL_MARK(0)
            __builtin_amdgcn_s_barrier();
            TL_MARK(1)
            const unsigned Pcb = Pc;
            for (int e = cb; e < ce; e++)
            {
                TL_PROBE_STEP
                const unsigned w = tp_uniform(wc_v);
                wc_v = s_word[min(e + 1, T - 1)];
                const unsigned ab = w & 0xFFu, bb = (w >> 8) & 0xFFu;
                const unsigned a4 = ((ab >> wr) & 1u) | (((ab >> (2 + wr)) & 1u) << 1) | (((ab >> (4 + wr)) & 1u) << 2) | (((ab >> (6 + wr)) & 1u) << 3);
                const unsigned b2 = ((bb >> wc) & 1u) | (((bb >> (wc + 4)) & 1u) << 1);
                const bool live = a4 && b2;
                const unsigned na = (unsigned)__builtin_popcount(ab);
#if PG_PLANES > 1
                const bool add = s_sign[w >> 20] < 0;
#define TP_MFMA(buf_)                                                                                                                       \
    _Pragma("unroll") for (int ni = 0; ni < 2; ni++)                                                                                       \
    {                                                                                                                                       \
        if (!((b2 >> ni) & 1u))                                                                                                             \
            continue;                                                                                                                       \
        _Pragma("unroll") for (int mi = 0; mi < 4; mi++) if ((a4 >> mi) & 1u)                                                              \
        {                                                                                                                                   \
            if (add)                                                                                                                        \
                acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[buf_][ni], fa[buf_][mi], acc[ni][mi], 0, 0, 0);                       \
            else                                                                                                                            \
                acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[buf_][ni], fa[buf_][mi], acc[ni][mi], 0, 0, DG_NEG_A);                \
        }                                                                                                                                   \
    }
#else
#define TP_MFMA(buf_)                                                                                                                       \
    _Pragma("unroll") for (int ni = 0; ni < 2; ni++)                                                                                       \
    {                                                                                                                                       \
        if (!((b2 >> ni) & 1u))                                                                                                             \
            continue;                                                                                                                       \
        _Pragma("unroll") for (int mi = 0; mi < 4; mi++) if ((a4 >> mi) & 1u)                                                              \
            acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[buf_][ni], fa[buf_][mi], acc[ni][mi], 0, 0, DG_NEG_A);                    \
    }
#endif
                // fragment addresses of this wavefront's pieces: slot = pieces before the step + live pieces below it, a scalar (dead
                // pieces point at some valid slot; their values are never used)
                unsigned va[4], vb[2];
                double fa[2][4], fb[2][2];
#define TP_READ(buf_, kq_)                                                                                                \
    {                                                                                                                     \
        _Pragma("unroll") for (int mi = 0; mi < 4; mi++) fa[buf_][mi] = *(tp_lds_cd)(unsigned long long)(va[mi] + (kq_) * 512u); \
        _Pragma("unroll") for (int ni = 0; ni < 2; ni++) fb[buf_][ni] = *(tp_lds_cd)(unsigned long long)(vb[ni] + ((kq_) >> 1) * 1024u + ((kq_) & 1) * 32u); \
    }
                if (live)
                {
                    nprod += (unsigned)(__builtin_popcount(a4) * __builtin_popcount(b2));
#pragma unroll
                    for (int ni = 0; ni < 2; ni++)
                        if ((b2 >> ni) & 1u)
                            touched |= a4 << (4 * ni);
#pragma unroll
                    for (int mi = 0; mi < 4; mi++)
                        va[mi] = a_frag + tp_scalar(((Pc + (unsigned)__builtin_popcount(ab & ((1u << (2 * mi + wr)) - 1u))) & (TP_RING - 1)) * TP_SLOT_BYTES);
#pragma unroll
                    for (int ni = 0; ni < 2; ni++)
                        vb[ni] = b_frag + tp_scalar(((Pc + na + (unsigned)__builtin_popcount(bb & ((1u << (wc + 4 * ni)) - 1u))) & (TP_RING - 1)) * TP_SLOT_BYTES);
                    TP_READ(0, 0)
                    TP_READ(1, 1)
                    TP_MFMA(0)
                }
                TL_MARK(2)
                // the following steps, behind this wavefront's first products (the slots of the batch before this one are free)
                if (e == cb)
                {
                    const unsigned keep = Pc;
                    Pc = Pcb; // (ring occupancy counts from the start of the batch being consumed)
                    try_issue();
                    Pc = keep;
                }
                TL_MARK(3)
                if (live)
                {
                    TP_READ(0, 2)
                    TP_MFMA(1)
                    TP_READ(1, 3)
                    TP_MFMA(0)
                    TP_MFMA(1)
                }
#undef TP_READ
#undef TP_MFMA
                Pc += na + (unsigned)__builtin_popcount(bb);
                TL_MARK(4)
            }
            cb = ce;
        }
        TL_PROBE_FLUSH
        TL_ITEM(4)
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
                        TP_C(ni, mi, r) = acc[ni][mi][r];
                }
        touched = 0; // (a subset of pre: everything has been written)
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
                        atomicAdd((double *)&TP_C(ni, mi, r), acc[ni][mi][r]);
            }
            continue;
        }
        double old[4][4];
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                old[mi][r] = ((t4 >> mi) & 1u) ? TP_C(ni, mi, r) : 0.0;
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
        {
            if (!((t4 >> mi) & 1u))
                continue;
#pragma unroll
            for (int r = 0; r < 4; r++)
                TP_C(ni, mi, r) = old[mi][r] + acc[ni][mi][r];
        }
    }
#ifdef TL_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TL_ITEM(5)
#endif
#undef TP_C
}
