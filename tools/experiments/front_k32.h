// front_k32.h -- EXPERIMENT (round 4): the dense-front update kernel with TWO slab steps per synchronisation.
// DESIGN.md 4.1 left two things untried; this is (a): K = 32 per barrier, one workgroup of sixteen wavefronts per CU (139 KB of LDS:
// two stages of a 128 x 32 A image + a 32 x 128 B image) instead of two workgroups of eight with K = 16.  Same DMA pipeline
// (global_load_lds_dwordx4, two stages, counted waits, bare s_barrier), same conflict-free images: A columns 144 doubles apart;
// B column n holds its 32 k's as 16 pairs, pair slot j ^ (n & 15) -- the XOR on the SOURCE side of the DMA.
// Wavefront w owns a 32 x 32 sub-tile (2 x 2 accumulators): rows (w & 3) * 32, columns (w >> 2) * 32.
// Included by tools/microbench/front_gemm.hip (which = 30000 + 100 * unit).
#pragma once

#define F32_KS 32
#define F32_A_DOUBLES (F32_KS * FR_LDA)
#define F32_STAGE_DOUBLES (F32_A_DOUBLES + FR_TILE * F32_KS) // 8704 doubles = 69 632 bytes
#define F32_THREADS 1024

__global__ __launch_bounds__(F32_THREADS, 1) void ssssm_front32_f64_kernel(const SsssmTaskD *__restrict__ tasks, int nb, const SsssmWorkD *__restrict__ work,
                                                                            unsigned long long *__restrict__ product_counter, unsigned unit)
{
    extern __shared__ __align__(16) double lds32[];
    const int tiles = nb / FR_TILE;
    const unsigned bid = logical_block_id(unit ? unit : (unsigned)(tiles * tiles));
    const SsssmWorkD G = work[bid];
    const int tile = (int)G.tile;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M0 = (tile % tiles) * FR_TILE, N0 = (tile / tiles) * FR_TILE;
    const int wm = (wave & 3) * 32, wn = (wave >> 2) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ntask = (int)(G.task_end - G.task_begin);
    const int spt = nb / F32_KS; // steps per task
    const int T = ntask * spt;
    const SsssmTaskD *my_tasks = tasks + G.task_begin;

    const unsigned a_voff = (unsigned)lane * 16u;
    const int bc = lane >> 4, bj = lane & 15; // B: lane = 16 c + j fetches pair slot j of column n = 4 g + c
    const unsigned b_voff = ((unsigned)bc * (unsigned)nb + 2u * (unsigned)(bj ^ ((4 * (wave & 3) + bc) & 15))) * 8u;

    auto issue = [&](int st)
    {
        const int t = st / spt, k0 = (st % spt) * F32_KS;
        const fr_gptr pa = (fr_gptr)reinterpret_cast<const char *>(my_tasks[t].a.val);
        const fr_gptr pb = (fr_gptr)reinterpret_cast<const char *>(my_tasks[t].b.val);
        double *stage = lds32 + (st & 1) * F32_STAGE_DOUBLES;
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
            const int k = wave + 16 * h;
            const fr_gptr src = dg_scalar_base(pa + ((size_t)(k0 + k) * nb + M0) * 8) + a_voff;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src, (fr_lptr)(stage + k * FR_LDA), 16, 0, 0);
        }
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
            const int g = wave + 16 * h; // columns 4 g .. 4 g + 3 (g & 3 = wave & 3)
            const fr_gptr src = dg_scalar_base(pb + ((size_t)(N0 + 4 * g) * nb + k0) * 8) + b_voff;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src, (fr_lptr)(stage + F32_A_DOUBLES + g * 128), 16, 0, 0);
        }
    };

    v4f64 acc[2][2];
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
#pragma unroll
        for (int mi = 0; mi < 2; mi++)
            acc[ni][mi] = (v4f64){0.0, 0.0, 0.0, 0.0};
    if (T > 0)
        issue(0);
    const int a_frag = l4 * FR_LDA + wm + l15;
    int b_frag[2];
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
        b_frag[ni] = F32_A_DOUBLES + (wn + ni * 16 + l15) * F32_KS + (l4 & 1);

    for (int st = 0; st < T; st++)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (st + 1 < T)
            issue(st + 1);
        const double *sA = lds32 + (st & 1) * F32_STAGE_DOUBLES;
        double fa[2][2], fb[2][2];
#define F32_READ(buf_, kq_)                                                                                   \
    {                                                                                                         \
        _Pragma("unroll") for (int mi = 0; mi < 2; mi++) fa[buf_][mi] = sA[a_frag + (kq_) * 4 * FR_LDA + mi * 16]; \
        _Pragma("unroll") for (int ni = 0; ni < 2; ni++) fb[buf_][ni] = sA[b_frag[ni] + 2 * ((2 * (kq_) + (l4 >> 1)) ^ l15)]; \
    }
        F32_READ(0, 0)
#pragma unroll
        for (int kq = 0; kq < F32_KS / 4; kq++)
        {
            if (kq + 1 < F32_KS / 4)
                F32_READ((kq + 1) & 1, kq + 1)
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int mi = 0; mi < 2; mi++)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[kq & 1][ni], fa[kq & 1][mi], acc[ni][mi], 0, 0, DG_NEG_A);
        }
#undef F32_READ
    }
    if (product_counter && lane == 0 && T)
        atomicAdd(product_counter, (unsigned long long)(8 * T)); // (two 16-deep products per piece and step, four pieces)

    double __attribute__((address_space(1))) *C = (double __attribute__((address_space(1))) *)reinterpret_cast<double *>(G.cdense);
    const unsigned c_voff = ((unsigned)l4 * (unsigned)nb + (unsigned)l15) * 8u;
#define F32_C(ni_, mi_, r_)                                                                          \
    (*(double __attribute__((address_space(1))) *)(dg_scalar_base((const char __attribute__((address_space(1))) *)C +                    \
                                                                  ((size_t)(N0 + wn + (ni_) * 16 + 4 * (r_)) * nb + M0 + wm) * 8) + \
                                                   dg_lane_offset(c_voff) + (mi_) * 128))
    double old[2][2][4];
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
#pragma unroll
        for (int mi = 0; mi < 2; mi++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                old[ni][mi][r] = F32_C(ni, mi, r);
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
#pragma unroll
        for (int mi = 0; mi < 2; mi++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                F32_C(ni, mi, r) = old[ni][mi][r] + acc[ni][mi][r];
#undef F32_C
}
