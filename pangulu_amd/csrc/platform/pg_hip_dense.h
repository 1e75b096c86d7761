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
// group in steps of 16: the 128 x 16 slab of A and the 16 x 128 slab of B go through LDS ([k][m] / [k][n], padded
// rows = conflict-free b64 fragment reads).  The product is formed transposed (A operand from the B slab, B operand
// from the A slab) so that accumulator register g of lane l is C(m0 + (l & 15), n0 + (l >> 4) + 4 g): the final
// read-modify-write of C moves 128-byte segments.
//
// Pipeline (round 2, third version).  Structural zeros come from the occupancy maps behind the mirrors: per window of
// DG_WINDOW tasks all 256 threads test one (task, K-slab) pair each and the live ones are compacted into a step list
// in LDS (one packed word per step: which 16-row pieces of A, which 16-column pieces of B, slab, task).  The main
// loop walks that list with TWO slabs of operands in flight in registers (sets X and Y) besides the one in LDS:
//     iteration j:   issue loads of step j+2  |  MFMAs of step j from LDS image j&1  |  wait for step j+1 only,
//                    write it to image (j+1)&1  |  barrier
// so a slab's loads have two product phases to arrive -- with partly filled blocks one phase (a few hundred to two
// thousand cycles) is shorter than a loaded memory round trip, and the wait at the LDS stage was 16 % of the kernel.
// Loads and stores share one counter (vmcnt) that retires in order, and the compiler's wait insertion assumes the worst
// at every join of control flow: the slab loads therefore are UNCONDITIONAL -- a piece that is structurally empty, and
// the two steps past the end of the list, load 16 bytes from the start of the mirror instead (an L1 hit, never used) --
// which keeps the count between "loads of step j+1" and "now" a compile-time constant (8).
// ---------------------------------------------------------------------------------------------------------------
#define DG_TILE 128
#define DG_K 16
// a workgroup-uniform address pinned in scalar registers: the compiler otherwise folds it into 64-bit per-lane arithmetic
// (base + lane offset first, then the uniform part on top, in vector instructions) instead of the scalar-base form of
// global_load / global_store (scalar base + 32-bit lane offset + immediate)
__device__ inline const char __attribute__((address_space(1))) *dg_scalar_base(const char __attribute__((address_space(1))) *p)
{
    unsigned long long v = (unsigned long long)p;
    asm("" : "+s"(v));
    return (const char __attribute__((address_space(1))) *)v;
}
// ... and the 32-bit lane offset kept as a value of the using basic block (hoisted out of it, its zero extension arrives as a
// 64-bit register pair and the scalar-base form is not selected)
__device__ inline unsigned dg_lane_offset(unsigned v)
{
    asm("" : "+v"(v));
    return v;
}
#ifndef DG_LD
// padded slab row (doubles).  144 (round 1) makes the MFMA fragment reads conflict-free but lands rows two apart on the same
// banks, which is exactly what the B-slab staging writes (eight threads write rows 0, 2, .. 14 of one column at once: an
// 8-way conflict per store); 146 / 148 spread those too: 24.6 -> 23.6 ms of update-kernel time (152: 24.8, 136: 24.2)
#define DG_LD 148
#endif
#ifndef DG_WINDOW
#define DG_WINDOW 16 // tasks whose bookkeeping is held in LDS at a time (one (task, K-slab) pair per thread: 16 or 32)
#endif
// f64 MFMAs reuse the BLGP immediate as NEG bits (bit 0: first source): D = C - A B without negating anything beforehand
#define DG_NEG_A 1

// Wavefronts per workgroup (compile time).
//   8 (default since the end of round 2): 64 x 32 sub-tiles (2 x 4 accumulators), 125 registers, two workgroups = four
//      wavefronts per SIMD; one slab in flight in registers, MFMA operand fragments single-buffered (no registers for
//      more -- the other three wavefronts of the SIMD cover those waits).
//   4: 64 x 64 sub-tiles (4 x 4 accumulators), 247 registers, two wavefronts per SIMD; two slabs in flight, fragments of the
//      next k-quarter read while the matrix cores work on the current one.
// Measured equal in the middle of round 2, when every slab still cost each wavefront ~100 vector instructions of address
// arithmetic (twice the wavefronts = twice that, against the matrix pipe: see dg_scalar_base).  After the diet: bench
// matrix 43.8-43.9 ms with 8 against 46.0-46.3 ms with 4 (same box, alternating runs), fem27(80) 166.9 against 170.3 ms.
#ifndef DG_WAVES
#define DG_WAVES 8
#endif
#define DG_THREADS (64 * DG_WAVES)
#define DG_NI (DG_WAVES == 8 ? 2 : 4) // 16-column pieces of C per wavefront (8 wavefronts: 64 x 32 sub-tiles, four wavefronts per SIMD)
#define DG_NST (16 / DG_WAVES)        // 16-byte pieces of each operand slab a thread stages (4 or 2)
#define DG_FB (DG_WAVES == 8 ? 1 : 2)  // MFMA operand fragment buffers
__global__ __launch_bounds__(DG_THREADS) __attribute__((amdgpu_waves_per_eu(DG_WAVES / 2, DG_WAVES / 2))) void ssssm_dense_f64_kernel(const SsssmTaskD *__restrict__ tasks, int nb,
                                                               unsigned long long *__restrict__ product_counter,
                                                               unsigned long long *dbg, const SsssmWorkD *__restrict__ work)
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
    // two LDS images of a slab pair: while the matrix cores consume one, the next slab is written into the other -- one
    // barrier per slab (75.8 KB per workgroup: two workgroups share a CU)
    __shared__ __align__(16) double sAb[2][DG_K * DG_LD];
    __shared__ __align__(16) double sBb[2][DG_K * DG_LD];
    const int tiles = nb / DG_TILE;
    const unsigned bid = logical_block_id((unsigned)(tiles * tiles)); // the tiles of one destination share operand halves: same XCD, same L2
    // destination tile and update queue of this workgroup: one self-contained item of the launch's work list (tiles no
    // update of the group can reach are left out).  Start-up is a chain of dependent memory round trips -- item, task
    // descriptors (with the operands' occupancy maps inside), first slabs -- and with ten live slabs per workgroup on
    // average (fem27(80)) every trip saved is a few percent of the kernel: round 2 had five (work index, group, tasks,
    // maps behind the mirrors, slabs).
    const SsssmWorkD G = work[bid];
    const int tile = (int)G.tile;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // (scalar: the skip tests below must be scalar branches)
    const int M0 = (tile % tiles) * DG_TILE, N0 = (tile / tiles) * DG_TILE; // workgroup tile origin
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * (16 * DG_NI);         // wavefront sub-tile inside it
    const int l15 = lane & 15, l4 = lane >> 4;

    v4f64 acc[DG_NI][4]; // [ni][mi]; initialised below, once it is known which pieces are going to be used

    // staging maps.  A slab (128 rows x 16 k, column-major source): thread -> rows 2*(tid & 63), +1 of k = (tid >> 6) + 4 i.
    // B slab (16 k x 128 cols): thread -> k pair 2*(tid & 7) of column (tid >> 3) + 32 i: 8 consecutive threads read one
    // 128-byte run of a column.
    const int a_m = 2 * (tid & 63), a_k = wave; // (a_k = tid >> 6, as a scalar)
    const int b_k = 2 * (tid & 7), b_n = tid >> 3;
    const int a_slab = (tid & 63) >> 3;                            // which 16-row slab of the tile this thread stages
    const int b_slab = __builtin_amdgcn_readfirstlane(tid >> 7);   // 16-column slab of piece i: b_slab + (DG_WAVES / 2) i
    // the two slabs in flight (first-class vector values: a struct type here is copied with memcpy through a stack slot),
    // loaded through global-address-space pointers (the mirror addresses pass through LDS as integers; left generic they
    // become flat loads, which also count against the LDS counter)
    typedef double v2f64 __attribute__((ext_vector_type(2)));
    typedef const v2f64 __attribute__((address_space(1))) *slab_ptr;
#if DG_WAVES == 8
    v2f64 raX[DG_NST], rbX[DG_NST];
#else
    v2f64 raX[DG_NST], rbX[DG_NST], raY[DG_NST], rbY[DG_NST];
#endif

    const u32 ntask = G.task_end - G.task_begin;
    const int steps_per_task = nb / DG_K; // <= 16: dense mode needs nb <= 256 (the occupancy maps have 16 x 16 bits)
    // For K-slab s of a task:  abits = which of the tile's eight 16-row slabs of A(:, s) hold pattern entries,
    //                          bbits = which of its eight 16-column slabs of B(s, :) do.
    // A slab is visited if both are non-zero; only live 16 x 16 pieces are fetched, and a wavefront issues the MFMAs
    // of a 16 x 16 x 16 product only when both its pieces are live -- for fill-in patterns of a few percent to a few
    // tens of percent that is a small fraction of the full contraction.  All of this is scalar (workgroup- or
    // wavefront-uniform) control flow.
    unsigned touched = 0, nprod = 0;
    // Per-task bookkeeping of up to DG_WINDOW tasks at a time lives in LDS, filled by all 256 threads at once (thread
    // = one (task, slab) pair): chasing task -> mirror -> map through global memory once per task costs microseconds
    // of exposed latency per task, more than the slabs of a sparse update themselves.
    __shared__ u32 s_step[DG_WINDOW * 16];   // live steps of the window in order: task << 20 | slab << 16 | bbits << 8 | abits
    __shared__ u32 s_cnt[DG_WAVES];
    __shared__ const double *s_pa[DG_WINDOW], *s_pb[DG_WINDOW];
#if PG_PLANES > 1
    __shared__ double s_sign[DG_WINDOW]; // sign of the task's product (complex updates as four real ones)
#define DG_SET_SIGN(t_, Tm_) s_sign[t_] = (Tm_).sign;
#else
#define DG_SET_SIGN(t_, Tm_)
#endif
    int win0 = -DG_WINDOW; // first task of the window in the tables
    int nwin = 0, iwin = 0; // steps in the window's list, next one to hand out

#define DG_FILL_WINDOW()                                                                             \
    {                                                                                                \
        __syncthreads(); /* nobody reads the previous window any more */                             \
        const int t_ = tid >> 4, s_ = tid & 15;                                                      \
        unsigned v_ = 0;                                                                             \
        if (tid < DG_WINDOW * 16 && win0 + t_ < (int)ntask && s_ < steps_per_task)                   \
        {                                                                                            \
            const SsssmTaskD &Tm_ = tasks[G.task_begin + win0 + t_];                                 \
            const double *pa_ = reinterpret_cast<const double *>(Tm_.a.val), *pb_ = reinterpret_cast<const double *>(Tm_.b.val); \
            unsigned ab_, bb_ = 0;                                                                   \
            if (Tm_.has_map)                                                                         \
            {                                                                                        \
                ab_ = ((unsigned)Tm_.amap[s_] >> (M0 / 16)) & 0xFFu;                                 \
                bb_ = ((unsigned)Tm_.bmap_t[s_] >> (N0 / 16)) & 0xFFu;                               \
            }                                                                                        \
            else                                                                                     \
            {                                                                                        \
                ab_ = ((unsigned)mirror_map(pa_, nb)[s_] >> (M0 / 16)) & 0xFFu;                      \
                const uint4 mb_ = *reinterpret_cast<const uint4 *>(mirror_map(pb_, nb) + N0 / 16);   \
                const unsigned w_[4] = {mb_.x, mb_.y, mb_.z, mb_.w};                                 \
                _Pragma("unroll") for (int c_ = 0; c_ < 8; c_++)                                     \
                    bb_ |= (((w_[c_ >> 1] >> (16 * (c_ & 1))) >> s_) & 1u) << c_;                    \
            }                                                                                        \
            if (ab_ && bb_ && (!G.slab_mask || ((G.slab_mask >> s_) & 1u)))                          \
                v_ = (bb_ << 8) | ab_ | ((unsigned)s_ << 16) | ((unsigned)t_ << 20);                 \
            if (s_ == 0)                                                                             \
            {                                                                                        \
                s_pa[t_] = pa_;                                                                      \
                s_pb[t_] = pb_;                                                                      \
                DG_SET_SIGN(t_, Tm_)                                                                 \
            }                                                                                        \
        }                                                                                            \
        const unsigned long long bal_ = __ballot(v_ != 0);                                           \
        if (lane == 0)                                                                               \
            s_cnt[wave] = (u32)__builtin_popcountll(bal_); /* (wavefronts 4..7: zero) */             \
        __syncthreads();                                                                             \
        unsigned at_ = (unsigned)__builtin_popcountll(bal_ & ((1ull << lane) - 1ull)), all_ = 0;     \
        _Pragma("unroll") for (int w_i = 0; w_i < DG_WAVES; w_i++)                                   \
        {                                                                                            \
            const unsigned c_ = s_cnt[w_i];                                                          \
            at_ += w_i < wave ? c_ : 0u;                                                             \
            all_ += c_;                                                                              \
        }                                                                                            \
        if (v_)                                                                                      \
            s_step[at_] = v_;                                                                        \
        __syncthreads();                                                                             \
        nwin = __builtin_amdgcn_readfirstlane((int)all_);                                            \
        iwin = 0;                                                                                    \
    }

    // next live step -> (ab, bb, k0, pa, pb, sign) as scalars; past the end: ab = bb = 0 and the destination's own mirror
    // as a harmless address for the dummy loads
#if PG_PLANES > 1
#define DG_STEP_SIGN(sg_, t_) sg_ = s_sign[t_];
#define DG_STEP_NOSIGN(sg_) sg_ = 1.0;
#else
#define DG_STEP_SIGN(sg_, t_)
#define DG_STEP_NOSIGN(sg_)
#endif
#define DG_FETCH(ab_, bb_, k0_, pa_, pb_, sg_)                                                       \
    {                                                                                                \
        while (iwin == nwin && win0 + DG_WINDOW < (int)ntask)                                        \
        {                                                                                            \
            win0 += DG_WINDOW;                                                                       \
            DG_FILL_WINDOW()                                                                         \
        }                                                                                            \
        if (iwin < nwin)                                                                             \
        {                                                                                            \
            const unsigned e_ = (unsigned)__builtin_amdgcn_readfirstlane((int)s_step[iwin]);         \
            iwin++;                                                                                  \
            ab_ = e_ & 0xFFu;                                                                        \
            bb_ = (e_ >> 8) & 0xFFu;                                                                 \
            k0_ = (int)((e_ >> 16) & 15u) * DG_K;                                                    \
            const unsigned long long a64_ = reinterpret_cast<unsigned long long>(s_pa[e_ >> 20]);    \
            const unsigned long long b64_ = reinterpret_cast<unsigned long long>(s_pb[e_ >> 20]);    \
            pa_ = reinterpret_cast<const double *>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a64_ >> 32)) << 32) | \
                                                   (unsigned)__builtin_amdgcn_readfirstlane((int)a64_)); \
            pb_ = reinterpret_cast<const double *>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b64_ >> 32)) << 32) | \
                                                   (unsigned)__builtin_amdgcn_readfirstlane((int)b64_)); \
            DG_STEP_SIGN(sg_, e_ >> 20)                                                              \
        }                                                                                            \
        else                                                                                         \
        {                                                                                            \
            ab_ = bb_ = 0;                                                                           \
            k0_ = 0;                                                                                 \
            pa_ = pb_ = reinterpret_cast<const double *>(G.cdense);                                  \
            DG_STEP_NOSIGN(sg_)                                                                      \
        }                                                                                            \
    }

    // all eight loads of a slab, unconditionally (see the header): dead pieces read 16 bytes that are in the mirror anyway.
    // Addresses are a workgroup-uniform base (scalar registers, scalar arithmetic) plus a per-thread byte offset that never
    // changes: an f64 MFMA occupies the SIMD's vector ALU for its whole 64 cycles -- no vector instruction of the other
    // wavefront on the SIMD issues beside it (tools/experiments/mfma_f64_coissue.hip: MFMA stream + integer stream on one
    // SIMD take the SUM of their times) -- so every vector instruction spent on address arithmetic is time taken from the
    // matrix pipe.  Round 2's first version computed 64-bit vector addresses per load (about 100 vector instructions per slab).
    typedef const char __attribute__((address_space(1))) *gbytes;
    const unsigned a_voff = (unsigned)a_m * 8u;                       // A: rows a_m, a_m+1 of column (k0 + a_k + 4 i), tile origin in the base
    const unsigned b_voff = ((unsigned)b_n * (unsigned)nb + (unsigned)b_k) * 8u; // B: rows k0 + b_k, +1 of column N0 + b_n + 32 i
    const unsigned a_bit = 1u << a_slab;
#define DG_LOAD_SLAB(ra_, rb_, ab_, bb_, k0_, pa_, pb_)                                              \
    {                                                                                                \
        const unsigned av_ = ((ab_) & a_bit) ? a_voff : 0u;                                          \
        const gbytes A_ = (gbytes)(pa_) + ((size_t)((k0_) + a_k) * nb + M0) * 8;                     \
        _Pragma("unroll") for (int i_ = 0; i_ < DG_NST; i_++)                                        \
            ra_[i_] = *(slab_ptr)(dg_scalar_base(A_ + (size_t)(DG_WAVES * i_) * nb * 8) + av_);      \
        _Pragma("unroll") for (int i_ = 0; i_ < DG_NST; i_++)                                        \
        {                                                                                            \
            const bool live_ = ((bb_) >> (b_slab + (DG_WAVES / 2) * i_)) & 1u;                       \
            const gbytes Bi_ = (gbytes)(pb_) + (live_ ? ((size_t)(N0 + 8 * DG_WAVES * i_) * nb + (k0_)) * 8 : (size_t)0); \
            rb_[i_] = *(slab_ptr)(dg_scalar_base(Bi_) + dg_lane_offset(live_ ? b_voff : 0u));      \
        }                                                                                            \
    }
#if PG_PLANES > 1
#define DG_SIGNED(v_, sg_) ((v_) * (sg_))
#else
#define DG_SIGNED(v_, sg_) (v_)
#endif
#define DG_STORE_LDS(buf_, ra_, rb_, sg_)                                                            \
    {                                                                                                \
        double *sA_ = sAb[buf_], *sB_ = sBb[buf_];                                                   \
        /* (pieces that were not fetched hold whatever the dummy load returned; no MFMA reads them) */ \
        _Pragma("unroll") for (int i_ = 0; i_ < DG_NST; i_++)                                        \
        {                                                                                            \
            *reinterpret_cast<v2f64 *>(&sA_[(a_k + DG_WAVES * i_) * DG_LD + a_m]) = DG_SIGNED(ra_[i_], sg_); \
            sB_[b_k * DG_LD + b_n + 8 * DG_WAVES * i_] = rb_[i_].x;                                  \
            sB_[(b_k + 1) * DG_LD + b_n + 8 * DG_WAVES * i_] = rb_[i_].y;                            \
        }                                                                                            \
    }
#define DG_PRODUCTS(buf_, ab_, bb_)                                                                  \
    {                                                                                                \
        const double *sA = sAb[buf_], *sB = sBb[buf_];                                               \
        const unsigned a4 = ((ab_) >> (wm / 16)) & 0xFu, b4 = ((bb_) >> (wn / 16)) & ((1u << DG_NI) - 1u); \
        if (a4 && b4)                                                                                \
        {                                                                                            \
            nprod += (unsigned)(__builtin_popcount(a4) * __builtin_popcount(b4));                    \
            _Pragma("unroll") for (int ni = 0; ni < DG_NI; ni++)                                     \
                if ((b4 >> ni) & 1u)                                                                 \
                    touched |= a4 << (4 * ni);                                                       \
            /* operand fragments of k-quarter kq+1 are read while the matrix cores work on kq: the LDS round trip      */ \
            /* (4 per slab) would otherwise sit in front of every group of sixteen MFMAs.  (8 wavefronts: no registers  */ \
            /* for that, and four wavefronts per SIMD to cover it: DG_FB = 1 buffer)                                    */ \
            double fa[DG_FB][4], fb[DG_FB][DG_NI];                                                   \
            if (DG_FB == 2)                                                                          \
            {                                                                                        \
                _Pragma("unroll") for (int mi = 0; mi < 4; mi++)                                     \
                    fa[0][mi] = sA[l4 * DG_LD + wm + mi * 16 + l15];                                 \
                _Pragma("unroll") for (int ni = 0; ni < DG_NI; ni++)                                 \
                    fb[0][ni] = sB[l4 * DG_LD + wn + ni * 16 + l15];                                 \
            }                                                                                        \
            _Pragma("unroll") for (int kq = 0; kq < DG_K / 4; kq++)                                  \
            {                                                                                        \
                const int kr = DG_FB == 2 ? kq + 1 : kq; /* the quarter read in this round */        \
                if (kr < DG_K / 4)                                                                   \
                {                                                                                    \
                    _Pragma("unroll") for (int mi = 0; mi < 4; mi++)                                 \
                        fa[kr % DG_FB][mi] = sA[(kr * 4 + l4) * DG_LD + wm + mi * 16 + l15];         \
                    _Pragma("unroll") for (int ni = 0; ni < DG_NI; ni++)                             \
                        fb[kr % DG_FB][ni] = sB[(kr * 4 + l4) * DG_LD + wn + ni * 16 + l15];         \
                }                                                                                    \
                _Pragma("unroll") for (int ni = 0; ni < DG_NI; ni++)                                 \
                {                                                                                    \
                    if (!((b4 >> ni) & 1u))                                                          \
                        continue;                                                                    \
                    _Pragma("unroll") for (int mi = 0; mi < 4; mi++)                                 \
                        if ((a4 >> mi) & 1u)                                                         \
                            acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[kq % DG_FB][ni], fa[kq % DG_FB][mi], acc[ni][mi], 0, 0, DG_NEG_A); \
                }                                                                                    \
            }                                                                                        \
        }                                                                                            \
    }

    // steps 0, 1, 2 ... alternate between the scalar sets (0: even steps, 1: odd steps)
    unsigned ab0, bb0, ab1, bb1;
    int k00, k01;
    const double *pa0, *pb0, *pa1, *pb1;
    double sg0 = 1.0, sg1 = 1.0;
    (void)sg0;
    (void)sg1;
    DG_FETCH(ab0, bb0, k00, pa0, pb0, sg0)
    DG_STAMP(0)
    if (!(ab0 | bb0))
    {
        if (stamping)
            atomicAdd(&dbg[7], 1ull << 32); // (empty workgroups in the high word)
        return; // nothing of these updates reaches this tile
    }
    DG_LOAD_SLAB(raX, rbX, ab0, bb0, k00, pa0, pb0)
    // A group that owns its destination and whose whole queue fits the bookkeeping window knows from the maps which
    // 16 x 16 pieces of C it is going to touch: their values go into the accumulators now, in flight together with the
    // first slabs (the MFMAs below subtract: acc = C - sum A B), and the epilogue is stores only -- read-modify-write at
    // the end costs memory round trips that nothing hides.
    unsigned pre = 0;
    // (CR64: one plane of the destination's mirror.)  Global address space spelled out: a pointer that comes out of a
    // descriptor in memory is generic to the compiler, and generic (flat) accesses also count against the LDS counter
    double __attribute__((address_space(1))) *C = (double __attribute__((address_space(1))) *)reinterpret_cast<double *>(G.cdense);
    // element (m, n) of this wavefront's piece (ni, mi), register r: scalar part (tile, wavefront, piece, register) + one
    // per-thread byte offset
    const unsigned c_voff = ((unsigned)l4 * (unsigned)nb + (unsigned)l15) * 8u;
#define DG_C(ni_, mi_, r_)                                                                           \
    (*(double __attribute__((address_space(1))) *)(dg_scalar_base((const char __attribute__((address_space(1))) *)C +                    \
                                                                  ((size_t)(N0 + wn + (ni_) * 16 + 4 * (r_)) * nb + M0 + wm) * 8) + \
                                                   dg_lane_offset(c_voff) + (mi_) * 128))
    if (ntask <= DG_WINDOW && !G.atomic)
    {
        unsigned m = 0;
        for (int e = lane; e < nwin; e += 64)
        {
            const unsigned v = s_step[e];
            const unsigned a4 = ((v & 0xFFu) >> (wm / 16)) & 0xFu, b4 = (((v >> 8) & 0xFFu) >> (wn / 16)) & ((1u << DG_NI) - 1u);
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
                        acc[ni][mi][r] = DG_C(ni, mi, r);
                }
                else
                {
                    // no product of this queue lands here and the epilogue leaves the piece alone: whatever the registers
                    // hold will do (not even the 8 moves that would zero them -- every vector instruction counts, see above)
                    asm volatile("" : "=v"(acc[ni][mi]));
                }
    }
    else
    {
#pragma unroll
        for (int ni = 0; ni < DG_NI; ni++)
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
                acc[ni][mi] = (v4f64){0.0, 0.0, 0.0, 0.0};
    }
#if DG_WAVES == 8
    // Eight wavefronts (128 registers each): ONE slab in flight -- set X only; four wavefronts per SIMD cover the rest.
    // Invariant at the loop head: image 0 holds step j (scalars 0), scalars 1 describe step j+1 (possibly past the end).
    DG_FETCH(ab1, bb1, k01, pa1, pb1, sg1)
    DG_STAMP(1)
    DG_STORE_LDS(0, raX, rbX, sg0)
    DG_STAMP(2)
    __syncthreads();
    DG_STAMP(3)
    do
    {
        DG_LOAD_SLAB(raX, rbX, ab1, bb1, k01, pa1, pb1)
        DG_STAMP(4)
        DG_PRODUCTS(0, ab0, bb0)
        DG_STAMP(5)
        DG_STORE_LDS(1, raX, rbX, sg1)
        DG_STAMP(2)
        __syncthreads();
        DG_STAMP(3)
        DG_FETCH(ab0, bb0, k00, pa0, pb0, sg0)
        DG_LOAD_SLAB(raX, rbX, ab0, bb0, k00, pa0, pb0)
        DG_STAMP(4)
        DG_PRODUCTS(1, ab1, bb1)
        DG_STAMP(5)
        if (stamping)
            atomicAdd(&dbg[7], (ab1 ? 2ull : 1ull)); // slab steps in the low word
        DG_STORE_LDS(0, raX, rbX, sg0)
        DG_STAMP(2)
        __syncthreads();
        DG_STAMP(3)
        DG_FETCH(ab1, bb1, k01, pa1, pb1, sg1)
    } while (ab0 | bb0);
#else
    DG_FETCH(ab1, bb1, k01, pa1, pb1, sg1)
    DG_LOAD_SLAB(raY, rbY, ab1, bb1, k01, pa1, pb1)
    DG_STAMP(1)
    DG_STORE_LDS(0, raX, rbX, sg0)
    DG_STAMP(2)
    __syncthreads();
    DG_STAMP(3)
    // No exit from the middle of the loop: a step past the end of the list is all dummy loads and no products, and the
    // single back edge keeps the compiler's view of what is in flight exact (with a break between the loads of one set
    // and their LDS stage it assumed them pending at the loop head and drained the whole queue there).
    do
    {
        // even step: image 0 holds it, set Y the next one; set X is free for the one after
        const unsigned cab0 = ab0, cbb0 = bb0;
        DG_FETCH(ab0, bb0, k00, pa0, pb0, sg0)
        DG_LOAD_SLAB(raX, rbX, ab0, bb0, k00, pa0, pb0)
        DG_STAMP(4)
        DG_PRODUCTS(0, cab0, cbb0)
        DG_STAMP(5)
        DG_STORE_LDS(1, raY, rbY, sg1) // (waits for the loads of set Y only: those of set X, issued above, stay in flight)
        DG_STAMP(2)
        __syncthreads(); // image 1 is complete, and everyone is done reading image 0
        DG_STAMP(3)
        // odd step, roles swapped
        const unsigned cab1 = ab1, cbb1 = bb1;
        DG_FETCH(ab1, bb1, k01, pa1, pb1, sg1)
        DG_LOAD_SLAB(raY, rbY, ab1, bb1, k01, pa1, pb1)
        DG_STAMP(4)
        DG_PRODUCTS(1, cab1, cbb1)
        DG_STAMP(5)
        if (stamping)
            atomicAdd(&dbg[7], (cab1 ? 2ull : 1ull)); // slab steps in the low word
        DG_STORE_LDS(0, raX, rbX, sg0)
        DG_STAMP(2)
        __syncthreads();
        DG_STAMP(3)
    } while (ab0 | bb0);
#endif
#undef DG_STORE_LDS
#undef DG_FETCH
#undef DG_FILL_WINDOW
#undef DG_LOAD_SLAB
#undef DG_PRODUCTS
#undef DG_SIGNED
#undef DG_SET_SIGN
#undef DG_STEP_SIGN
#undef DG_STEP_NOSIGN
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
                        DG_C(ni, mi, r) = acc[ni][mi][r];
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
                    if (acc[ni][mi][r] != 0.0)
                        atomicAdd((double *)&DG_C(ni, mi, r), acc[ni][mi][r]);
                }
            }
            continue;
        }
        double old[4][4];
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                old[mi][r] = ((t4 >> mi) & 1u) ? DG_C(ni, mi, r) : 0.0;
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
        {
            if (!((t4 >> mi) & 1u))
                continue; // no product reached this 16 x 16 piece of C
#pragma unroll
            for (int r = 0; r < 4; r++)
                DG_C(ni, mi, r) = old[mi][r] + acc[ni][mi][r];
        }
    }
    DG_STAMP(6)
#undef DG_STAMP
#undef DG_C
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
    return val_t{(real_t)dense[at], (real_t)dense[at + mirror_plane_stride(nb)]};
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
