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

    // ---- DMA side: per-lane source offsets (bytes)
    const unsigned a_voff = (unsigned)lane * 16u;
    const int a_piece = lane >> 3;
    const int bc = lane >> 3, bj = lane & 7;
    const unsigned b_voff = ((unsigned)bc * (unsigned)nb + 2u * (unsigned)(bj ^ ((4 * (wave & 1) + (bc >> 1)) & 7))) * 8u;
    const unsigned a_col_off = (unsigned)wave * (unsigned)nb * 8u;     // this wavefront's A column of every slab
    const unsigned b_grp_off = (unsigned)(8 * wave) * (unsigned)nb * 8u; // this wavefront's eight B columns

    // ---- issue cursor: walks the items of this workgroup and their step records; `ic_rec` is the record of the NEXT step to request
    unsigned ic_k = 0;             // ordinal of the item the cursor is in
    unsigned ic_left = 0;          // steps of that item not yet requested
    const SsssmStepD *ic_ptr = nullptr;
    bool ic_done = false;
    auto ic_enter = [&](unsigned k)
    {
        const unsigned it = pos + k * G_;
        ic_left = SW_CONST(SsssmItemInfoD, info + it)->nsteps;
        ic_ptr = steps + (SW_CONST(SsssmWorkD, work + it)->pad_ >> 1);
    };
    // (the record of the cursor's next step is fetched a step ahead -- scalar loads, consumed at the next request: their latency
    //  never sits between a barrier and the products behind it)
    unsigned long long rec_pa = 0, rec_pb = 0;
    unsigned rec_w = 0;
    auto fetch_rec = [&]()
    {
        const SsssmStepD __attribute__((address_space(4))) *R = SW_CONST(SsssmStepD, ic_ptr);
        rec_pa = R->pa;
        rec_pb = R->pb;
        // (this wavefront's nibble of the step's masks rides in the word: what the compute side needs is one scalar register per step)
        rec_w = R->word | (((unsigned)(R->masks >> (4 * wave)) & 0xFu) << SW_NIBBLE_SHIFT) | SW_REAL;
    };
    ic_enter(0);
    fetch_rec();
    // words of the steps in flight: q0 = the step being computed, q1..q3 = requested
    unsigned q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    const unsigned lds0 = (unsigned)(unsigned long long)(fr_lptr)sw_lds; // (the workgroup's dynamic LDS starts here)
    const unsigned dma_a = lds0 + (unsigned)wave * (FR_LDA * 8u), dma_b = lds0 + (FR_KS * FR_LDA + (unsigned)wave * 128u) * 8u;
    auto request = [&](unsigned stage_bytes) -> unsigned
    {
        // requests the slab of the cursor's next step into the stage at `stage_bytes`; returns its word (0 and no request when the
        // stream has ended)
        if (ic_done)
            return 0u;
        const unsigned w = rec_w;
        const fr_gptr pa = (fr_gptr)rec_pa, pb = (fr_gptr)rec_pb;
        const unsigned ab = w & 0xFFu, bb = (w >> 8) & 0xFFu;
        {
            const bool a_live = (ab >> a_piece) & 1u;
            const unsigned off = a_live ? a_col_off + a_voff : 0u;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(dg_scalar_base(pa) + off), (fr_lptr)(unsigned long long)(dma_a + stage_bytes), 16, 0, 0);
        }
        {
            const bool b_live = (bb >> (wave >> 1)) & 1u;
            const unsigned off = b_live ? b_grp_off + b_voff : 0u;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(dg_scalar_base(pb) + off), (fr_lptr)(unsigned long long)(dma_b + stage_bytes), 16, 0, 0);
        }
        ic_ptr++;
        if (--ic_left == 0)
        {
            if (++ic_k < nmine)
                ic_enter(ic_k);
            else
                ic_done = true;
        }
        if (!ic_done)
            fetch_rec();
        return w;
    };

    // ---- compute side
    const int a_frag = l4 * FR_LDA + pi * 16 + l15; // + kq * 4 * FR_LDA + mi * 64
    const int swz = (l15 >> 1) & 7;
    const int b_frag = FR_KS * FR_LDA + (pj * 16 + l15) * 16 + (l4 & 1); // + ni * 64 * 16 + 2 * ((2 kq + (l4 >> 1)) ^ swz)
    const unsigned c_voff = ((unsigned)l4 * (unsigned)nb + (unsigned)l15) * 8u;
    // piece (ni, mi) register r of lane l: C(M0 + (pi + 4 mi) * 16 + (l & 15), N0 + (pj + 4 ni) * 16 + 4 r + (l >> 4))
#define SW_C(base_, M0_, N0_, ni_, mi_, r_)                                                                                               \
    (*(double __attribute__((address_space(1))) *)(dg_scalar_base((const char __attribute__((address_space(1))) *)(base_) +              \
                                                                  ((size_t)((N0_) + (pj + 4 * (ni_)) * 16 + 4 * (r_)) * nb + (M0_) + pi * 16) * 8) + \
                                                   dg_lane_offset(c_voff) + (mi_) * 512))
    // Two accumulator sets, used in turn: the item being computed lives in one, the pieces of the item behind it arrive in the other
    // (no register moves at a boundary, and no copies the compiler could make of registers whose loads are still in flight).
    v4f64 acc0[2][2], acc1[2][2]; // [ni][mi]
    auto item_fields = [&](unsigned k, double *&cbase, int &M0, int &N0, unsigned &pre, bool &atomic)
    {
        const unsigned it = pos + k * G_;
        const SsssmWorkD __attribute__((address_space(4))) *W = SW_CONST(SsssmWorkD, work + it);
        cbase = (double *)(unsigned long long)W->cdense;
        const unsigned tile = W->tile;
        M0 = (int)(tile % (unsigned)tiles) * FR_TILE;
        N0 = (int)(tile / (unsigned)tiles) * FR_TILE;
        atomic = W->atomic != 0;
        pre = (unsigned)(SW_CONST(SsssmItemInfoD, info + it)->pre >> (4 * wave)) & 0xFu;
    };
    double *c_cur, *c_nxt = nullptr;
    int M0c, N0c, M0n = 0, N0n = 0;
    unsigned pre_c, pre_n = 0;
    bool at_c, at_n = false;
    item_fields(0, c_cur, M0c, N0c, pre_c, at_c);
    // the destination pieces this wavefront will touch go into the accumulators (the matrix cores subtract: acc = C - sum A B);
    // an item that ADDS its sum with atomics (a queue cut along K) starts from zero
#define SW_LOAD_PIECES(dst_, base_, M0_, N0_, pre_, atomic_)                                                      \
    _Pragma("unroll") for (int ni = 0; ni < 2; ni++) _Pragma("unroll") for (int mi = 0; mi < 2; mi++)             \
    {                                                                                                               \
        if ((((pre_) >> (mi + 2 * ni)) & 1u) && !(atomic_))                                                         \
        {                                                                                                           \
            _Pragma("unroll") for (int r = 0; r < 4; r++) dst_[ni][mi][r] = SW_C(base_, M0_, N0_, ni, mi, r);       \
        }                                                                                                           \
        else                                                                                                        \
            dst_[ni][mi] = (v4f64){0.0, 0.0, 0.0, 0.0};                                                             \
    }
    // The same at an item boundary, for the item after the next -- as inline assembly: the compiler's wait insertion, which cannot
    // follow the two accumulator sets through the loops, otherwise puts s_waitcnt vmcnt(0) in front of EVERY product (measured on the
    // ISA: 32 of 32).  These loads are older than every slab request behind them and loads return in order, so the first counted
    // wait two steps later covers them; the set is not touched before the boundary of the item in between (vmcnt(0)).
#define SW_LOAD_PIECES_ASYNC(dst_, base_, M0_, N0_, pre_, atomic_)                                                                        \
    _Pragma("unroll") for (int ni = 0; ni < 2; ni++) _Pragma("unroll") for (int mi = 0; mi < 2; mi++)                                     \
    {                                                                                                                                     \
        if ((((pre_) >> (mi + 2 * ni)) & 1u) && !(atomic_))                                                                               \
        {                                                                                                                                 \
            _Pragma("unroll") for (int r = 0; r < 4; r++)                                                                                 \
            {                                                                                                                             \
                const unsigned long long sb_ = (unsigned long long)(base_) + ((size_t)((N0_) + (pj + 4 * ni) * 16 + 4 * r) * nb + (M0_) + pi * 16) * 8; \
                if (mi == 0)                                                                                                              \
                    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst_[ni][mi][r]) : "v"(c_voff), "s"(sb_) : "memory");            \
                else                                                                                                                      \
                    asm volatile("global_load_dwordx2 %0, %1, %2 offset:512" : "=v"(dst_[ni][mi][r]) : "v"(c_voff), "s"(sb_) : "memory"); \
            }                                                                                                                             \
        }                                                                                                                                 \
        else                                                                                                                              \
            dst_[ni][mi] = (v4f64){0.0, 0.0, 0.0, 0.0};                                                                                   \
    }
    SW_LOAD_PIECES(acc0, c_cur, M0c, N0c, pre_c, at_c)
    if (nmine > 1)
    {
        item_fields(1, c_nxt, M0n, N0n, pre_n, at_n);
        SW_LOAD_PIECES(acc1, c_nxt, M0n, N0n, pre_n, at_n)
    }
    unsigned cm_k = 0; // ordinal of the item being computed

    // waits as builtins, not inline assembly: the compiler's own wait insertion sees them (s_waitcnt immediates of gfx9: vmcnt in
    // bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at "no wait")
#define SW_WAIT_VM(n_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n_) & 0xF) | (((n_) >> 4) << 14))
    // ---- prologue: three slabs requested, the first one landed everywhere, its fragments of the first two k-quarters read
    const unsigned STB = (unsigned)(FR_STAGE_DOUBLES * 8);
    q0 = request(0u);
    q1 = request(STB);
    q2 = request(2u * STB);
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
    // byte addresses inside stage 0: A fragments at a_byte (+ kq * 4 columns, + 64 rows), B fragments at b_byte[kq] (+ 64 columns)
    const unsigned a_byte = lds0 + (unsigned)a_frag * 8u;
    unsigned b_byte[4];
#pragma unroll
    for (int kq = 0; kq < 4; kq++)
        b_byte[kq] = lds0 + (unsigned)(b_frag + 2 * ((2 * kq + (l4 >> 1)) ^ swz)) * 8u;
    // all four fragments of k-quarter kq of the stage at byte offset so_ into buffer buf_
#define SW_READ_ALL(buf_, so_, kq_)                                                 \
    {                                                                               \
        const unsigned aa_ = a_byte + (so_), bb_ = b_byte[kq_] + (so_);             \
        SW_LDS_READ(fa[buf_][0], aa_, (kq_) * 4 * FR_LDA * 8);                      \
        SW_LDS_READ(fa[buf_][1], aa_, (kq_) * 4 * FR_LDA * 8 + 512);                \
        SW_LDS_READ(fb[buf_][0], bb_, 0);                                           \
        SW_LDS_READ(fb[buf_][1], bb_, 8192);                                        \
    }
    unsigned nprod = 0;
#ifdef SW_PROBE // (tools/microbench/front_gemm.hip: where does a step spend its cycles?  wavefronts 0 and 5 of every workgroup)
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
    int skip_waits = 0;     // steps behind an item boundary whose slabs had landed there
    unsigned so_cur = 0;    // byte offset of the stage of the step being computed
    // The critical section of a step -- between the last product a wavefront issues and the first one of the next step, when all sixteen
    // wavefronts do the same thing and nothing feeds the matrix pipes -- holds a wait, the barrier and a wait: the fragments of the
    // first two k-quarters were read during the previous step, the masks are in registers, and ALL bookkeeping (the slab request, the
    // cursor, the queue of words, stage offsets) sits between the products of the first and the second k-quarter, where the scalar
    // instructions of one wavefront issue beside the products of the other three of its SIMD.
#if PG_PLANES > 1
#define SW_ONE_PRODUCT(buf_, ni_, mi_)                                                                                              \
    if ((m4 >> ((mi_) + 2 * (ni_))) & 1u)                                                                                           \
    {                                                                                                                               \
        if (add)                                                                                                                    \
            acc[ni_][mi_] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[buf_][ni_], fa[buf_][mi_], acc[ni_][mi_], 0, 0, 0);             \
        else                                                                                                                        \
            acc[ni_][mi_] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[buf_][ni_], fa[buf_][mi_], acc[ni_][mi_], 0, 0, DG_NEG_A);      \
    }
#else
#define SW_ONE_PRODUCT(buf_, ni_, mi_)                                                                                              \
    if ((m4 >> ((mi_) + 2 * (ni_))) & 1u)                                                                                           \
        acc[ni_][mi_] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[buf_][ni_], fa[buf_][mi_], acc[ni_][mi_], 0, 0, DG_NEG_A);
#endif
#define SW_PRODUCTS(buf_)                \
    {                                    \
        asm volatile("" : "+s"(m4));     \
        SW_ONE_PRODUCT(buf_, 0, 0)       \
        SW_ONE_PRODUCT(buf_, 0, 1)       \
        SW_ONE_PRODUCT(buf_, 1, 0)       \
        SW_ONE_PRODUCT(buf_, 1, 1)       \
    }
    // one item: its steps on the accumulator set `acc`; at its end the set is stored and takes the pieces of the item after the next.
    // Returns true when that was this workgroup's last item.
    auto run_item = [&](v4f64(&acc)[2][2]) -> bool
    {
        for (;;)
        {
            // ---- top of step c: slab c has landed (this wavefront's two requests by the wait, everybody's behind the barrier);
            //      everybody has finished with slab c - 1
            SW_MARK(5)
            if (skip_waits > 0)
                skip_waits--;
            else if (q2 & SW_REAL)
                SW_WAIT_VM(4); // (the requests of slabs c + 1 and c + 2 may stay in flight)
            else if (q1 & SW_REAL)
                SW_WAIT_VM(2);
            else
                SW_WAIT_VM(0);
            SW_MARK(0)
            __builtin_amdgcn_s_barrier();
            SW_MARK(1)
            unsigned m4 = (q0 >> SW_NIBBLE_SHIFT) & 0xFu; // this wavefront's live pieces in this step
#if PG_PLANES > 1
            const bool add = (q0 & SW_ADD) != 0;
#endif
            const unsigned so_this = so_cur;
            const bool last = (q0 & SW_LAST) != 0;
            if (m4)
            {
                SW_READ_ALL(0, so_this, 0)
                SW_READ_ALL(1, so_this, 1)
                SW_FRAGS_READY(0, 4);
                SW_MARK(2)
                SW_PRODUCTS(0)
            }
            SW_MARK(3)
            // ---- bookkeeping, in the shadow of the first products: slab c + 3 into the stage slab c - 1 was in, the queue of words
            {
                const unsigned so_req = so_cur == 0u ? 3u * STB : so_cur - STB;
                q3 = request(so_req);
                nprod += (unsigned)__builtin_popcount(m4);
                q0 = q1;
                q1 = q2;
                q2 = q3;
                so_cur = so_cur + STB == 4u * STB ? 0u : so_cur + STB;
            }
            SW_MARK(4)
#ifdef SW_PROBE
            pr_sum[7]++;
#endif
            if (m4)
            {
                SW_READ_ALL(0, so_this, 2)
                SW_FRAGS_READY(1, 4);
                SW_PRODUCTS(1)
                SW_READ_ALL(1, so_this, 3)
                SW_FRAGS_READY(0, 4);
                SW_PRODUCTS(0)
                SW_FRAGS_READY(1, 0);
                SW_PRODUCTS(1)
            }
            if (!last)
                continue;
            // ---- item boundary: everything requested so far lands first (the newest request is a step old), then the stores
            SW_WAIT_VM(0);
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int mi = 0; mi < 2; mi++)
                    if ((pre_c >> (mi + 2 * ni)) & 1u)
                    {
                        if (at_c)
                        {
#pragma unroll
                            for (int r = 0; r < 4; r++)
                                if (acc[ni][mi][r] != 0.0)
                                    atomicAdd((double *)&SW_C(c_cur, M0c, N0c, ni, mi, r), acc[ni][mi][r]);
                        }
                        else
                        {
#pragma unroll
                            for (int r = 0; r < 4; r++)
                                SW_C(c_cur, M0c, N0c, ni, mi, r) = acc[ni][mi][r];
                        }
                    }
            if (++cm_k >= nmine)
                return true;
            c_cur = c_nxt;
            M0c = M0n;
            N0c = N0n;
            pre_c = pre_n;
            at_c = at_n;
            if (cm_k + 1 < nmine)
            {
                item_fields(cm_k + 1, c_nxt, M0n, N0n, pre_n, at_n);
                SW_LOAD_PIECES_ASYNC(acc, c_nxt, M0n, N0n, pre_n, at_n)
            }
            skip_waits = 3; // (slabs c + 1 .. c + 3 had landed at the boundary)
            return false;
        }
    };
    for (;;)
    {
        if (run_item(acc0))
            break;
        if (run_item(acc1))
            break;
    }
#undef SW_READ_ALL
#undef SW_LDS_READ
#undef SW_FRAGS_READY
#undef SW_ONE_PRODUCT
#undef SW_PRODUCTS
#undef SW_WAIT_VM
#undef SW_LOAD_PIECES
#undef SW_LOAD_PIECES_ASYNC
#undef SW_C
    if (product_counter && lane == 0 && nprod)
        atomicAdd(product_counter, (unsigned long long)nprod);
#ifdef SW_PROBE
    if (lane == 0 && (wave == 0 || wave == 5))
        for (int i_ = 0; i_ < 8; i_++)
            atomicAdd(&g_sw_probe[i_], pr_sum[i_]);
#endif
#undef SW_MARK
}
