// pg_hip_stream.h -- ssssm_stream_f64_kernel: the general MFMA update kernel as ONE persistent workgroup of sixteen wavefronts per
// CU that walks a STREAM of slab steps across work items (round 6).
// (included by pg_hip_platform.hip after pg_hip_front.h; tools/microbench/front_gemm.hip checks and times it stand-alone)
//
// Why.  ssssm_tilesv_f64_kernel (pg_hip_front.h) runs a work item -- a 128 x 128 destination tile and its queue of updates -- per
// workgroup of eight wavefronts, two workgroups per CU.  Its probes and counters: MFMA pipes
// 61 % busy; a step-pair costs 2 W_busiest + F with F = 2000 cycles nothing overlaps and the busiest wavefront at 1.3 x the mean; a
// work item's life outside its step loop (descriptor chain, step-list bookkeeping behind three barriers, destination preload,
// epilogue) is 20-48 % of it.  On gfx950 an f64 MFMA occupies its SIMD's vector pipe, so every vector instruction of the bookkeeping
// (ballots, readfirstlanes of step records, address arithmetic) is paid in matrix-pipe time.
//
// What changes.
//  * The step lists are built by a small kernel of their own in front of the update kernel (ssssm_stream_build_kernel: the
//    occupancy tests, the compaction, the per-wavefront masks of touched pieces) into scratch memory: 32 bytes per live (task,
//    K-slab) step with the operand addresses of THAT slab of THAT tile folded in.  The update kernel reads them with scalar loads:
//    no LDS step list, no readfirstlane, no descriptor chain in the update kernel at all.
//  * One workgroup of sixteen wavefronts per CU owns the whole LDS: FOUR stages of the 128 x 16 + 16 x 128 slab images, slabs
//    requested three steps ahead by global_load_lds (two DMA instructions per wavefront and step: one A column, one half B piece).
//    The wait before the barrier of step c covers slab c + 1, so the fragments of the first k-quarter of step c + 1 are read BEFORE
//    its barrier, behind the products of step c: after a barrier the matrix cores start at once.
//  * A wavefront owns FOUR pieces of the tile -- rows {i, i + 4} x columns {j, j + 4}, the pairs (i, j) spread over the SIMDs as a
//    Latin square ((i + j) mod 4 = SIMD), so a rectangle of live pieces loads the four matrix pipes evenly -- : 32 accumulator
//    registers instead of 64, which leaves room for the NEXT item's destination pieces: they are loaded while the current item
//    computes, and an item ends with stores and 32 register moves.  The workgroup is persistent over its share of the launch's
//    items (static, round-robin in the host's heavy-first order with the tiles of a destination on one XCD), so the pipeline never
//    drains between items: the issue cursor simply walks on into the next item's steps.
//  * vmcnt: loads return in order, stores do not (with respect to loads).  A step's wait is vmcnt(2) -- everything but the newest
//    slab request -- which is safe whatever stores are in flight (they only make it wait longer).  An item boundary waits for
//    everything BEFORE it issues its stores (cheap: the newest request is a step old), and the two steps behind it skip their
//    waits (their slabs had landed at the boundary) instead of waiting behind the stores.
#pragma once

#define SW_THREADS 1024
#define SW_WAVES 16
#define SW_STAGES 4
#define SW_LDS_BYTES (SW_STAGES * FR_STAGE_DOUBLES * 8)
#define SW_LAST 0x80000000u // step word: last step of its item
#define SW_REAL 0x40000000u // (in the kernel's queue of words: a step, not the end of the stream)
#define SW_ADD 0x01000000u  // step word: the product is ADDED (complex updates: the A_im B_im product on the real plane)
#define SW_NIBBLE_SHIFT 20  // (in the kernel's queue of words: bits 20-23 = this wavefront's nibble of `masks`)

struct SsssmStepD // one live (task, K-slab) step of one tile
{
    unsigned long long pa; // A mirror + (k0 * nb + M0) doubles: column k of the slab at + k * nb, the tile's 128 rows from there
    unsigned long long pb; // B mirror + (N0 * nb + k0) doubles: column n of the tile at + n * nb, the slab's 16 rows from there
    u32 word;              // ab (bits 0-7: live 16-row pieces of A in the tile), bb (8-15: live 16-column pieces of B), SW_ADD, SW_LAST
    u32 pad_;
    unsigned long long masks; // nibble w: which of wavefront w's four pieces have a live product in this step (sw_wave_mask)
};
static_assert(sizeof(SsssmStepD) == 32, "one s_load_dwordx8 per step");

struct SsssmItemInfoD // what the list builder found out about a work item
{
    u32 nsteps;             // live steps (at least 1: an item without any gets one step without live pieces)
    u32 pad_;
    unsigned long long pre; // nibble w: which of wavefront w's four pieces some step of the queue touches
};

// Scratch written by an EARLIER kernel, read here through the constant address space: loads with a uniform address then are scalar
// loads (s_load) whatever stores this kernel makes -- through a global pointer the compiler has to assume its own stores may clobber
// them and turns them into vector loads, which would count in vmcnt and break the counted waits.
#define SW_CONST(T, p) ((const T __attribute__((address_space(4))) *)(unsigned long long)(p))

// wavefront w = 4 a + s (s = its SIMD): row pieces {a, a + 4}, column pieces {j, j + 4} with j = (s - a) mod 4
__host__ __device__ inline unsigned sw_wave_mask(unsigned w, unsigned ab, unsigned bb)
{
    const unsigned i = w >> 2, j = ((w & 3u) - i) & 3u;
    const unsigned r0 = (ab >> i) & 1u, r1 = (ab >> (i + 4)) & 1u, c0 = (bb >> j) & 1u, c1 = (bb >> (j + 4)) & 1u;
    return (r0 & c0) | ((r1 & c0) << 1) | ((r0 & c1) << 2) | ((r1 & c1) << 3); // bit mi + 2 ni
}

// ---------------------------------------------------------------------------------------------------------------
// The list builder: one workgroup of 256 threads per work item.  work[].pad_ = (first step slot << 1) | all_live.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ssssm_stream_build_kernel(const SsssmTaskD *__restrict__ tasks, int nb, const SsssmWorkD *__restrict__ work,
                                                                 SsssmItemInfoD *__restrict__ info, SsssmStepD *__restrict__ steps)
{
    __shared__ u32 s_cnt[4];
    __shared__ unsigned long long s_pre;
    const SsssmWorkD G = work[blockIdx.x];
    const int tiles = nb / FR_TILE, tile = (int)G.tile;
    const int M0 = (tile % tiles) * FR_TILE, N0 = (tile / tiles) * FR_TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntask = (int)(G.task_end - G.task_begin), nslab = nb / FR_KS;
    const bool all_live = (G.pad_ & 1u) != 0;
    SsssmStepD *out = steps + (G.pad_ >> 1);
    const SsssmTaskD *my_tasks = tasks + G.task_begin;
    if (tid == 0)
        s_pre = 0;
    __syncthreads();
    unsigned total = 0;
    unsigned long long pre = 0;
    for (int win0 = 0; win0 < ntask; win0 += 16)
    {
        unsigned v = 0;
        unsigned long long pa_v = 0, pb_v = 0, step_masks = 0;
        const int t_ = tid >> 4, s_ = tid & 15;
        if (win0 + t_ < ntask && s_ < nslab)
        {
            const SsssmTaskD &Tm = my_tasks[win0 + t_];
            const double *pa_ = reinterpret_cast<const double *>(Tm.a.val), *pb_ = reinterpret_cast<const double *>(Tm.b.val);
            unsigned ab_, bb_ = 0;
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
            {
                v = (bb_ << 8) | ab_ | 0x40000000u; // (bit 30: a live step, whatever the masks)
#if PG_PLANES > 1
                if (Tm.sign < 0)
                    v |= SW_ADD;
#endif
                pa_v = (unsigned long long)(pa_ + ((size_t)(s_ * FR_KS) * nb + M0));
                pb_v = (unsigned long long)(pb_ + ((size_t)N0 * nb + s_ * FR_KS));
#pragma unroll
                for (unsigned w = 0; w < SW_WAVES; w++)
                    step_masks |= (unsigned long long)sw_wave_mask(w, ab_, bb_) << (4 * w);
                pre |= step_masks;
            }
        }
        const unsigned long long bal = __ballot(v != 0);
        if (lane == 0)
            s_cnt[wave] = (u32)__builtin_popcountll(bal);
        __syncthreads();
        unsigned at = (unsigned)__builtin_popcountll(bal & ((1ull << lane) - 1ull)), all = 0;
#pragma unroll
        for (int w_i = 0; w_i < 4; w_i++)
        {
            const unsigned c_ = s_cnt[w_i];
            at += w_i < wave ? c_ : 0u;
            all += c_;
        }
        if (v)
        {
            SsssmStepD r;
            r.pa = pa_v;
            r.pb = pb_v;
            r.word = v & ~0x40000000u;
            r.pad_ = 0;
            r.masks = step_masks;
            out[total + at] = r;
        }
        total += all;
        __syncthreads();
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        pre |= (unsigned long long)__shfl_xor((long long)pre, off, 64);
    if (lane == 0 && pre)
        atomicOr(&s_pre, pre);
    __syncthreads();
    if (tid == 0)
    {
        if (total == 0)
        {
            // (nothing live: one step without live pieces keeps the update kernel's cursors simple -- its requests read the start of
            //  the destination, its products are none, nothing is stored)
            SsssmStepD r;
            r.pa = r.pb = (unsigned long long)reinterpret_cast<double *>(G.cdense);
            r.word = SW_LAST;
            r.pad_ = 0;
            r.masks = 0;
            out[0] = r;
            total = 1;
        }
        else
            out[total - 1].word |= SW_LAST;
        SsssmItemInfoD I;
        I.nsteps = total;
        I.pad_ = 0;
        I.pre = s_pre;
        info[blockIdx.x] = I;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The update kernel.  grid = G workgroups (G a multiple of 32, at most the number of CUs), G <= ceil32(items).
// Position p of a round of G items -> workgroup: XCD (p / 4) mod 8, slot (p / 32) * 4 + p mod 4, i.e. the four tiles of a
// destination run on one XCD at about the same time (they share operand halves: one L2).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SW_THREADS) void ssssm_stream_f64_kernel(int nb, const SsssmWorkD *__restrict__ work, const SsssmItemInfoD *__restrict__ info,
                                                                      const SsssmStepD *__restrict__ steps, unsigned nitems,
                                                                      unsigned long long *__restrict__ product_counter)
{
    extern __shared__ __align__(16) double sw_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int pi = wave >> 2, pj = ((wave & 3) - pi) & 3; // row pieces pi + 4 mi, column pieces pj + 4 ni
    const int tiles = nb / FR_TILE;
    const unsigned G_ = gridDim.x, bx = blockIdx.x;
    const unsigned xcd = bx & 7u, slot = bx >> 3;
    const unsigned pos = (((slot >> 2) << 3) + xcd) * 4u + (slot & 3u); // this workgroup's position in every round of G_ items
    if (pos >= nitems)
        return;
    const unsigned nmine = (nitems - pos + G_ - 1) / G_; // items pos, pos + G_, ...

    // ---- DMA side.  Per-lane source offsets (bytes) as CONSTANT vector registers: on gfx950 an f64 matrix product occupies its SIMD's
    // vector pipe, and a vector instruction issued beside the products of three other wavefronts waits for a slot behind them -- the
    // cycle probes of the first version found 763 cycles per step in a request whose offsets took six vector instructions, and the
    // youngest wavefront of a SIMD, which runs last, pays that serially.  So: no vector arithmetic in the step loop at all.  Lanes
    // whose 16-row piece of A is dead are switched off in EXEC for the one instruction (the lane mask = the piece bits, each
    // replicated eight times: three s_bitreplicate); a dead half piece of B reads through a zero offset register instead.
    const unsigned a_voff = (unsigned)lane * 16u;
    const int bc = lane >> 3, bj = lane & 7;
    const unsigned b_voff = ((unsigned)bc * (unsigned)nb + 2u * (unsigned)(bj ^ ((4 * (wave & 1) + (bc >> 1)) & 7))) * 8u;
    unsigned a_off_vec = (unsigned)wave * (unsigned)nb * 8u + a_voff;       // this wavefront's A column of every slab
    unsigned b_off_vec = (unsigned)(8 * wave) * (unsigned)nb * 8u + b_voff; // this wavefront's eight B columns
    unsigned v_zero = 0u;
    asm volatile("" : "+v"(a_off_vec), "+v"(b_off_vec), "+v"(v_zero)); // (kept in registers, not rematerialised inside the loop)
    const unsigned b_bit = 8u + ((unsigned)wave >> 1);                   // the bit of the step word that says this wavefront's B half piece is live

    // ---- issue cursor: walks the items of this workgroup and their step records; rec_* is the record of the NEXT step to request
    unsigned ic_k = 0;    // ordinal of the item the cursor is in
    unsigned ic_left = 0; // steps of that item not yet requested
    const SsssmStepD *ic_ptr = nullptr;
    bool ic_done = false;
    unsigned long long rec_pa = 0, rec_pb = 0;
    unsigned rec_w = 0;
#define SW_IC_ENTER(k_)                                                                  \
    {                                                                                    \
        const unsigned it_ = pos + (k_) * G_;                                            \
        ic_left = SW_CONST(SsssmItemInfoD, info + it_)->nsteps;                          \
        ic_ptr = steps + (SW_CONST(SsssmWorkD, work + it_)->pad_ >> 1);                  \
    }
    // (the record is fetched a step ahead -- scalar loads, consumed at the next request; this wavefront's nibble of the step's masks
    //  rides in the word: what the compute side needs is one scalar register per step)
#define SW_FETCH_REC                                                                                                    \
    {                                                                                                                   \
        const SsssmStepD __attribute__((address_space(4))) *R_ = SW_CONST(SsssmStepD, ic_ptr);                          \
        rec_pa = R_->pa;                                                                                                \
        rec_pb = R_->pb;                                                                                                \
        rec_w = R_->word | (((unsigned)(R_->masks >> (4 * wave)) & 0xFu) << SW_NIBBLE_SHIFT) | SW_REAL;                 \
    }
    SW_IC_ENTER(0u)
    SW_FETCH_REC
    const unsigned lds0 = (unsigned)(unsigned long long)(fr_lptr)sw_lds; // (the workgroup's dynamic LDS starts here)
    const unsigned STB = (unsigned)(FR_STAGE_DOUBLES * 8);
    const unsigned dma_a = lds0 + (unsigned)wave * (FR_LDA * 8u), dma_b = lds0 + (FR_KS * FR_LDA + (unsigned)wave * 128u) * 8u;
    // one DMA instruction: 64 lanes x 16 bytes from sbase + voff to the LDS address in M0 (+ 16 bytes per lane)
#define SW_DMA(voff_, sbase_, m0_) \
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff_), "s"(sbase_), "s"(m0_) : "memory")
#define SW_DMA_LANES(voff_, sbase_, m0_, lanes_) \
    asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, %3\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(voff_), "s"(sbase_), "s"(m0_), "s"(lanes_) : "memory")
#define SW_BITREP(dst_, src_) asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(dst_) : "s"(src_))
    // requests the slab of the cursor's next step into stage T_ (compile time) and puts its word into Q_ (0 and no request when the
    // stream has ended: the waits at the top of a step look at the words)
#define SW_REQUEST(T_, Q_)                                                                          \
    if (ic_done)                                                                                    \
        Q_ = 0u;                                                                                    \
    else                                                                                            \
    {                                                                                               \
        const unsigned w_ = rec_w;                                                                  \
        if ((w_ & 0xFFu) == 0xFFu)                                                                  \
            SW_DMA(a_off_vec, rec_pa, dma_a + (T_) * STB);                                          \
        else                                                                                        \
        {                                                                                           \
            unsigned long long l1_, l2_, l3_;                                                       \
            SW_BITREP(l1_, w_ & 0xFFu);                                                             \
            SW_BITREP(l2_, (unsigned)l1_);                                                          \
            SW_BITREP(l3_, (unsigned)l2_);                                                          \
            SW_DMA_LANES(a_off_vec, rec_pa, dma_a + (T_) * STB, l3_);                               \
        }                                                                                           \
        if ((w_ >> b_bit) & 1u)                                                                     \
            SW_DMA(b_off_vec, rec_pb, dma_b + (T_) * STB);                                          \
        else                                                                                        \
            SW_DMA(v_zero, rec_pb, dma_b + (T_) * STB);                                             \
        ic_ptr++;                                                                                   \
        if (--ic_left == 0)                                                                         \
        {                                                                                           \
            if (++ic_k < nmine)                                                                     \
                SW_IC_ENTER(ic_k)                                                                   \
            else                                                                                    \
                ic_done = true;                                                                     \
        }                                                                                           \
        if (!ic_done)                                                                               \
            SW_FETCH_REC                                                                            \
        Q_ = w_;                                                                                    \
    }

    // ---- compute side
    const int a_frag = l4 * FR_LDA + pi * 16 + l15; // + kq * 4 * FR_LDA + mi * 64
    const int swz = (l15 >> 1) & 7;
    const int b_frag = FR_KS * FR_LDA + (pj * 16 + l15) * 16 + (l4 & 1); // + ni * 64 * 16 + 2 * ((2 kq + (l4 >> 1)) ^ swz)
    // piece (ni, mi) register r of lane l: C(M0 + (pi + 4 mi) * 16 + (l & 15), N0 + (pj + 4 ni) * 16 + 4 r + (l >> 4)).  Addresses:
    // a scalar base per item and ni (the piece's first column, the tile's rows), a CONSTANT lane offset register per r
    // (column 4 r + (l >> 4), row l & 15), mi = + 512 bytes as an immediate -- nothing to compute per access.
    unsigned c_vr[4];
#pragma unroll
    for (int r = 0; r < 4; r++)
        c_vr[r] = ((unsigned)(4 * r + l4) * (unsigned)nb + (unsigned)l15) * 8u;
    asm volatile("" : "+v"(c_vr[0]), "+v"(c_vr[1]), "+v"(c_vr[2]), "+v"(c_vr[3]));
#define SW_CBASE(base_, M0_, N0_, ni_) ((unsigned long long)(base_) + ((size_t)((N0_) + (pj + 4 * (ni_)) * 16) * nb + (M0_) + pi * 16) * 8)
    // acc = the item being computed; nxt = the pieces of the item behind it, arriving while this one computes (inline-assembly loads:
    // the compiler's wait insertion, which cannot know that they are covered by the counted waits of the step loop, otherwise puts
    // s_waitcnt vmcnt(0) in front of EVERY product -- measured on the ISA: 32 of 32).  At a boundary: stores, then 32 register moves.
    v4f64 acc[2][2], nxt[2][2]; // [ni][mi]
#define SW_ITEM_FIELDS(k_, cbase_, M0_, N0_, pre_, atomic_)                                                   \
    {                                                                                                         \
        const unsigned it_ = pos + (k_) * G_;                                                                 \
        const SsssmWorkD __attribute__((address_space(4))) *W_ = SW_CONST(SsssmWorkD, work + it_);            \
        cbase_ = (double *)(unsigned long long)W_->cdense;                                                    \
        const unsigned tile_ = W_->tile;                                                                      \
        M0_ = (int)(tile_ % (unsigned)tiles) * FR_TILE;                                                       \
        N0_ = (int)(tile_ / (unsigned)tiles) * FR_TILE;                                                       \
        atomic_ = W_->atomic != 0;                                                                            \
        pre_ = (unsigned)(SW_CONST(SsssmItemInfoD, info + it_)->pre >> (4 * wave)) & 0xFu;                    \
    }
    double *c_cur, *c_nxt = nullptr;
    int M0c, N0c, M0n = 0, N0n = 0;
    unsigned pre_c, pre_n = 0;
    bool at_c, at_n = false;
    SW_ITEM_FIELDS(0u, c_cur, M0c, N0c, pre_c, at_c)
    // the destination pieces this wavefront will touch go into the accumulators (the matrix cores subtract: acc = C - sum A B);
    // an item that ADDS its sum with atomics (a queue cut along K) starts from zero.  These loads are older than every slab request
    // behind them and loads return in order, so the counted waits of the step loop cover them; nxt is not read before a boundary
    // (vmcnt(0)).
#define SW_LOAD_PIECES_ASYNC(dst_, base_, M0_, N0_, pre_, atomic_)                                                                        \
    _Pragma("unroll") for (int ni = 0; ni < 2; ni++)                                                                                      \
    {                                                                                                                                     \
        const unsigned long long sb_ = SW_CBASE(base_, M0_, N0_, ni);                                                                     \
        _Pragma("unroll") for (int mi = 0; mi < 2; mi++)                                                                                  \
        {                                                                                                                                 \
            if ((((pre_) >> (mi + 2 * ni)) & 1u) && !(atomic_))                                                                           \
            {                                                                                                                             \
                _Pragma("unroll") for (int r = 0; r < 4; r++)                                                                             \
                {                                                                                                                         \
                    if (mi == 0)                                                                                                          \
                        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst_[ni][mi][r]) : "v"(c_vr[r]), "s"(sb_) : "memory");       \
                    else                                                                                                                  \
                        asm volatile("global_load_dwordx2 %0, %1, %2 offset:512" : "=v"(dst_[ni][mi][r]) : "v"(c_vr[r]), "s"(sb_) : "memory"); \
                }                                                                                                                         \
            }                                                                                                                             \
            else                                                                                                                          \
                dst_[ni][mi] = (v4f64){0.0, 0.0, 0.0, 0.0};                                                                               \
        }                                                                                                                                 \
    }
#define SW_STORE_PIECES(src_, base_, M0_, N0_, pre_, atomic_)                                                                             \
    _Pragma("unroll") for (int ni = 0; ni < 2; ni++)                                                                                      \
    {                                                                                                                                     \
        const unsigned long long sb_ = SW_CBASE(base_, M0_, N0_, ni);                                                                     \
        _Pragma("unroll") for (int mi = 0; mi < 2; mi++) if (((pre_) >> (mi + 2 * ni)) & 1u)                                              \
        {                                                                                                                                 \
            if (atomic_)                                                                                                                  \
            {                                                                                                                             \
                _Pragma("unroll") for (int r = 0; r < 4; r++) if (src_[ni][mi][r] != 0.0)                                                 \
                    atomicAdd((double *)(sb_ + c_vr[r] + mi * 512), src_[ni][mi][r]);                                                     \
            }                                                                                                                             \
            else                                                                                                                          \
            {                                                                                                                             \
                _Pragma("unroll") for (int r = 0; r < 4; r++)                                                                             \
                {                                                                                                                         \
                    if (mi == 0)                                                                                                          \
                        asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(c_vr[r]), "v"(src_[ni][mi][r]), "s"(sb_) : "memory");        \
                    else                                                                                                                  \
                        asm volatile("global_store_dwordx2 %0, %1, %2 offset:512" ::"v"(c_vr[r]), "v"(src_[ni][mi][r]), "s"(sb_) : "memory"); \
                }                                                                                                                         \
            }                                                                                                                             \
        }                                                                                                                                 \
    }
    SW_LOAD_PIECES_ASYNC(acc, c_cur, M0c, N0c, pre_c, at_c)
    if (nmine > 1)
    {
        SW_ITEM_FIELDS(1u, c_nxt, M0n, N0n, pre_n, at_n)
        SW_LOAD_PIECES_ASYNC(nxt, c_nxt, M0n, N0n, pre_n, at_n)
    }
    else
    {
#pragma unroll
        for (int ni = 0; ni < 2; ni++)
#pragma unroll
            for (int mi = 0; mi < 2; mi++)
                nxt[ni][mi] = (v4f64){0.0, 0.0, 0.0, 0.0};
    }
    unsigned cm_k = 0; // ordinal of the item being computed

    // waits as builtins, not inline assembly: the compiler's own wait insertion sees them (s_waitcnt immediates of gfx9: vmcnt in
    // bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at "no wait")
#define SW_WAIT_VM(n_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n_) & 0xF) | (((n_) >> 4) << 14))
    // words of the steps in flight, by STAGE: the step loop is unrolled over the four stages, so that stage addresses are immediates
    // and the queue needs no moves
    unsigned q_0 = 0, q_1 = 0, q_2 = 0, q_3 = 0;
    // ---- prologue: three slabs requested and landed everywhere
    SW_REQUEST(0u, q_0)
    SW_REQUEST(1u, q_1)
    SW_REQUEST(2u, q_2)
    SW_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    double fa[2][2], fb[2][2]; // [buffer][mi] / [buffer][ni]
    // Fragment reads as inline assembly: the compiler's wait insertion treats every LDS read it knows of as a possible reader of every
    // LDS-DMA in flight and puts s_waitcnt vmcnt(0) in front of the use -- which would serialise the four-stage pipeline (the two-stage
    // kernels never noticed: their waits are vmcnt(0) anyway).  The waits for these reads are SW_FRAGS_READY below; "+v" ties the
    // products behind it.
#define SW_LDS_READ(dst_, addr_, off_) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "n"(off_) : "memory")
    // (LDS reads return in order: with `newer_` reads issued behind the ones a product needs, lgkmcnt(newer_) says those have arrived --
    //  a scalar load in flight only makes the wait stricter)
#define SW_FRAGS_READY(buf_, newer_) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fa[buf_][0]), "+v"(fa[buf_][1]), "+v"(fb[buf_][0]), "+v"(fb[buf_][1]) : "n"(newer_) : "memory")
    // fragment addresses: one register per operand and k-quarter for stages 0 / 1 and one for stages 2 / 3; the stage inside the pair,
    // the k-quarter of A and the second piece are immediates (< 64 KiB)
    unsigned fa_lo = lds0 + (unsigned)a_frag * 8u, fa_hi = fa_lo + 2u * STB;
    unsigned fb_lo[4], fb_hi[4];
#pragma unroll
    for (int kq = 0; kq < 4; kq++)
    {
        fb_lo[kq] = lds0 + (unsigned)(b_frag + 2 * ((2 * kq + (l4 >> 1)) ^ swz)) * 8u;
        fb_hi[kq] = fb_lo[kq] + 2u * STB;
    }
    asm volatile("" : "+v"(fa_lo), "+v"(fa_hi), "+v"(fb_lo[0]), "+v"(fb_lo[1]), "+v"(fb_lo[2]), "+v"(fb_lo[3]), "+v"(fb_hi[0]), "+v"(fb_hi[1]), "+v"(fb_hi[2]), "+v"(fb_hi[3]));
    // all four fragments of k-quarter kq_ of stage S_ into buffer buf_
#define SW_READ_ALL(buf_, S_, kq_)                                                                                        \
    {                                                                                                                     \
        SW_LDS_READ(fa[buf_][0], ((S_) < 2 ? fa_lo : fa_hi), ((S_)&1) * (FR_STAGE_DOUBLES * 8) + (kq_) * 4 * FR_LDA * 8);      \
        SW_LDS_READ(fa[buf_][1], ((S_) < 2 ? fa_lo : fa_hi), ((S_)&1) * (FR_STAGE_DOUBLES * 8) + (kq_) * 4 * FR_LDA * 8 + 512); \
        SW_LDS_READ(fb[buf_][0], ((S_) < 2 ? fb_lo[kq_] : fb_hi[kq_]), ((S_)&1) * (FR_STAGE_DOUBLES * 8));                     \
        SW_LDS_READ(fb[buf_][1], ((S_) < 2 ? fb_lo[kq_] : fb_hi[kq_]), ((S_)&1) * (FR_STAGE_DOUBLES * 8) + 8192);              \
    }
    unsigned nprod = 0;
#ifdef SW_PROBE // (tools/microbench/front_gemm.hip: where does a step spend its cycles?  wavefronts 0 and 15 of every workgroup)
    unsigned long long pr_t = __builtin_readcyclecounter(), pr_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SW_MARK(i)                                                  \
    {                                                               \
        const unsigned long long n_ = __builtin_readcyclecounter(); \
        pr_sum[i] += n_ - pr_t;                                     \
        pr_t = n_;                                                  \
    }
#else
#define SW_MARK(i)
#endif
    int skip_waits = 0; // steps behind an item boundary whose slabs had landed there
    unsigned m4 = 0;    // this wavefront's live pieces in the step being computed
#if PG_PLANES > 1
    bool add = false;
#define SW_ONE_PRODUCT(acc_, buf_, ni_, mi_)                                                                                        \
    if ((m4 >> ((mi_) + 2 * (ni_))) & 1u)                                                                                           \
    {                                                                                                                               \
        if (add)                                                                                                                    \
            acc_[ni_][mi_] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[buf_][ni_], fa[buf_][mi_], acc_[ni_][mi_], 0, 0, 0);           \
        else                                                                                                                        \
            acc_[ni_][mi_] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[buf_][ni_], fa[buf_][mi_], acc_[ni_][mi_], 0, 0, DG_NEG_A);    \
    }
#else
#define SW_ONE_PRODUCT(acc_, buf_, ni_, mi_)                                                                                        \
    if ((m4 >> ((mi_) + 2 * (ni_))) & 1u)                                                                                           \
        acc_[ni_][mi_] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[buf_][ni_], fa[buf_][mi_], acc_[ni_][mi_], 0, 0, DG_NEG_A);
#endif
#define SW_PRODUCTS(acc_, buf_)            \
    {                                      \
        asm volatile("" : "+s"(m4));       \
        SW_ONE_PRODUCT(acc_, buf_, 0, 0)   \
        SW_ONE_PRODUCT(acc_, buf_, 0, 1)   \
        SW_ONE_PRODUCT(acc_, buf_, 1, 0)   \
        SW_ONE_PRODUCT(acc_, buf_, 1, 1)   \
    }
    // An item boundary: everything requested so far lands first (the newest request is most of a step old; nxt's loads are older),
    // then the stores; acc takes nxt, and nxt's loads for the item after the next go out.
#define SW_BOUNDARY                                                                                                  \
    {                                                                                                                \
        SW_WAIT_VM(0);                                                                                               \
        SW_STORE_PIECES(acc, c_cur, M0c, N0c, pre_c, at_c)                                                           \
        if (++cm_k >= nmine)                                                                                         \
            goto sw_done;                                                                                            \
        _Pragma("unroll") for (int ni = 0; ni < 2; ni++) _Pragma("unroll") for (int mi = 0; mi < 2; mi++) acc[ni][mi] = nxt[ni][mi]; \
        c_cur = c_nxt;                                                                                               \
        M0c = M0n;                                                                                                   \
        N0c = N0n;                                                                                                   \
        pre_c = pre_n;                                                                                               \
        at_c = at_n;                                                                                                 \
        if (cm_k + 1 < nmine)                                                                                        \
        {                                                                                                            \
            SW_ITEM_FIELDS(cm_k + 1, c_nxt, M0n, N0n, pre_n, at_n)                                                   \
            SW_LOAD_PIECES_ASYNC(nxt, c_nxt, M0n, N0n, pre_n, at_n)                                                  \
        }                                                                                                            \
        skip_waits = 3; /* (slabs c + 1 .. c + 3 had landed at the boundary) */                                      \
    }
    // One step on stage S_ (compile time); its word is Q0_, the two behind it Q1_ and Q2_, and the request of slab c + 3 goes into
    // the stage and the word slot slab c - 1 had (T_, Q3_).  ALL bookkeeping sits behind the first k-quarter's products: between the
    // last product a wavefront issues and the first one of its next step there is a wait, the barrier, eight fragment reads, a wait.
#define SW_STEP(S_, T_, Q0_, Q1_, Q2_, Q3_)                                                                         \
    {                                                                                                               \
        SW_MARK(5)                                                                                                  \
        if (skip_waits > 0)                                                                                         \
            skip_waits--;                                                                                           \
        else if (Q2_ & SW_REAL)                                                                                     \
            SW_WAIT_VM(4); /* (the requests of slabs c + 1 and c + 2 may stay in flight) */                         \
        else if (Q1_ & SW_REAL)                                                                                     \
            SW_WAIT_VM(2);                                                                                          \
        else                                                                                                        \
            SW_WAIT_VM(0);                                                                                          \
        SW_MARK(0)                                                                                                  \
        __builtin_amdgcn_s_barrier();                                                                               \
        SW_MARK(1)                                                                                                  \
        m4 = (Q0_ >> SW_NIBBLE_SHIFT) & 0xFu;                                                                       \
        SW_ADD_FLAG(Q0_)                                                                                            \
        if (m4)                                                                                                     \
        {                                                                                                           \
            SW_READ_ALL(0, S_, 0)                                                                                   \
            SW_READ_ALL(1, S_, 1)                                                                                   \
            SW_FRAGS_READY(0, 4);                                                                                   \
            SW_MARK(2)                                                                                              \
            SW_PRODUCTS(acc, 0)                                                                                     \
        }                                                                                                           \
        SW_MARK(3)                                                                                                  \
        const bool last_ = (Q0_ & SW_LAST) != 0;                                                                    \
        nprod += (unsigned)__builtin_popcount(m4);                                                                  \
        SW_REQUEST(T_, Q3_)                                                                                         \
        SW_MARK(4)                                                                                                  \
        SW_PROBE_COUNT                                                                                              \
        if (m4)                                                                                                     \
        {                                                                                                           \
            SW_READ_ALL(0, S_, 2)                                                                                   \
            SW_FRAGS_READY(1, 4);                                                                                   \
            SW_PRODUCTS(acc, 1)                                                                                     \
            SW_READ_ALL(1, S_, 3)                                                                                   \
            SW_FRAGS_READY(0, 4);                                                                                   \
            SW_PRODUCTS(acc, 0)                                                                                     \
            SW_FRAGS_READY(1, 0);                                                                                   \
            SW_PRODUCTS(acc, 1)                                                                                     \
        }                                                                                                           \
        if (last_)                                                                                                  \
            SW_BOUNDARY                                                                                             \
    }
#if PG_PLANES > 1
#define SW_ADD_FLAG(Q_) add = ((Q_)&SW_ADD) != 0;
#else
#define SW_ADD_FLAG(Q_)
#endif
#ifdef SW_PROBE
#define SW_PROBE_COUNT pr_sum[7]++;
#else
#define SW_PROBE_COUNT
#endif
    // the stream: stage = step mod 4, whatever the item
    for (;;)
    {
        SW_STEP(0, 3u, q_0, q_1, q_2, q_3)
        SW_STEP(1, 0u, q_1, q_2, q_3, q_0)
        SW_STEP(2, 1u, q_2, q_3, q_0, q_1)
        SW_STEP(3, 2u, q_3, q_0, q_1, q_2)
    }
sw_done:;
#undef SW_STEP
#undef SW_BOUNDARY
#undef SW_ADD_FLAG
#undef SW_PROBE_COUNT
#undef SW_READ_ALL
#undef SW_LDS_READ
#undef SW_FRAGS_READY
#undef SW_ONE_PRODUCT
#undef SW_PRODUCTS
#undef SW_WAIT_VM
#undef SW_LOAD_PIECES_ASYNC
#undef SW_ITEM_FIELDS
#undef SW_REQUEST
#undef SW_DMA
#undef SW_DMA_LANES
#undef SW_BITREP
#undef SW_IC_ENTER
#undef SW_FETCH_REC
#undef SW_CBASE
#undef SW_STORE_PIECES
    if (product_counter && lane == 0 && nprod)
        atomicAdd(product_counter, (unsigned long long)nprod);
#ifdef SW_PROBE
    if (lane == 0 && (wave == 0 || wave == 15))
        for (int i_ = 0; i_ < 8; i_++)
            atomicAdd(&g_sw_probe[i_], pr_sum[i_]);
#endif
#undef SW_MARK
}
