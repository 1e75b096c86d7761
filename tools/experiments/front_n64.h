// front_n64.h -- EXPERIMENT (round 4): the dense-front update kernel on 128 x 64 tiles with FOUR wavefronts per workgroup.
// The K = 32 experiment (front_k32.h) said that one workgroup of sixteen wavefronts per CU is slower than two of eight: what
// hides a workgroup's per-step gap (wait, barrier, DMA issue, first fragments) is OTHER workgroups on the same SIMDs.  So: more,
// smaller workgroups -- 128 x 64 tiles, four wavefronts (one per SIMD, 64 x 32 each as before), two stages of a 128 x 16 A image +
// a 16 x 64 B image = 53 KB of LDS: three workgroups per CU, each SIMD holds wavefronts of three independent barrier domains.
// Operand traffic per flop grows by a half (the A slab serves 64 columns instead of 128).
// Included by tools/microbench/front_gemm.hip (which = 40000 + 100 * unit): grid = 2 x work items, workgroup 2 i + h = columns 64 h.. of item i.
#pragma once

#define F64N_TILE_N 64
#define F64N_STAGE_DOUBLES (FR_KS * FR_LDA + F64N_TILE_N * FR_KS) // 2304 + 1024 = 3328 doubles = 26 624 bytes
#define F64N_THREADS 256

__global__ __launch_bounds__(F64N_THREADS, 3) void ssssm_front_n64_f64_kernel(const SsssmTaskD *__restrict__ tasks, int nb, const SsssmWorkD *__restrict__ work,
                                                                               unsigned long long *__restrict__ product_counter, unsigned unit)
{
    __shared__ __align__(16) double lds[2 * F64N_STAGE_DOUBLES];
    const int tiles = nb / FR_TILE;
    const unsigned bid2 = logical_block_id(unit ? 2 * unit : (unsigned)(2 * tiles * tiles));
    const SsssmWorkD G = work[bid2 >> 1];
    const int tile = (int)G.tile;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M0 = (tile % tiles) * FR_TILE, N0 = (tile / tiles) * FR_TILE + (int)(bid2 & 1u) * F64N_TILE_N;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ntask = (int)(G.task_end - G.task_begin);
    const int steps_shift = nb == 256 ? 4 : 3;
    const int steps_per_task = 1 << steps_shift;
    const int T = ntask << steps_shift;
    const SsssmTaskD *my_tasks = tasks + G.task_begin;

    const unsigned a_voff = (unsigned)lane * 16u;
    const int bc = lane >> 3, bj = lane & 7;
    unsigned b_voff[2];
#pragma unroll
    for (int par = 0; par < 2; par++)
        b_voff[par] = ((unsigned)bc * (unsigned)nb + 2u * (unsigned)(bj ^ ((4 * par + (bc >> 1)) & 7))) * 8u;

    auto issue = [&](int st)
    {
        // six DMA instructions per wave: A columns wave, wave + 4, + 8, + 12; B column groups wave, wave + 4 (eight columns each)
        const int t = st >> steps_shift, k0 = (st & (steps_per_task - 1)) * FR_KS;
        const fr_gptr pa = (fr_gptr)reinterpret_cast<const char *>(my_tasks[t].a.val);
        const fr_gptr pb = (fr_gptr)reinterpret_cast<const char *>(my_tasks[t].b.val);
        double *stage = lds + (st & 1) * F64N_STAGE_DOUBLES;
#pragma unroll
        for (int h = 0; h < 4; h++)
        {
            const int k = wave + 4 * h;
            const fr_gptr src = dg_scalar_base(pa + ((size_t)(k0 + k) * nb + M0) * 8) + a_voff;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src, (fr_lptr)(stage + k * FR_LDA), 16, 0, 0);
        }
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
            const int g = wave + 4 * h; // (g & 1 = wave & 1)
            const fr_gptr src = dg_scalar_base(pb + ((size_t)(N0 + 8 * g) * nb + k0) * 8) + b_voff[wave & 1];
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src, (fr_lptr)(stage + FR_KS * FR_LDA + g * 128), 16, 0, 0);
        }
    };

    v4f64 acc[2][4];
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
            acc[ni][mi] = (v4f64){0.0, 0.0, 0.0, 0.0};
    if (T > 0)
        issue(0);
    const int a_frag = l4 * FR_LDA + wm + l15;
    int b_frag[2][2];
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
    {
        const int n = wn + ni * 16 + l15;
        b_frag[ni][0] = FR_KS * FR_LDA + n * 16 + (l4 & 1);
        b_frag[ni][1] = (n >> 1) & 7;
    }
    for (int st = 0; st < T; st++)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (st + 1 < T)
            issue(st + 1);
        const double *sA = lds + (st & 1) * F64N_STAGE_DOUBLES;
        double fa[2][4], fb[2][2];
#define F64N_READ(buf_, kq_)                                                                                  \
    {                                                                                                         \
        _Pragma("unroll") for (int mi = 0; mi < 4; mi++) fa[buf_][mi] = sA[a_frag + (kq_) * 4 * FR_LDA + mi * 16]; \
        _Pragma("unroll") for (int ni = 0; ni < 2; ni++) fb[buf_][ni] = sA[b_frag[ni][0] + 2 * ((2 * (kq_) + (l4 >> 1)) ^ b_frag[ni][1])]; \
    }
        F64N_READ(0, 0)
#pragma unroll
        for (int kq = 0; kq < FR_KS / 4; kq++)
        {
            if (kq + 1 < FR_KS / 4)
                F64N_READ((kq + 1) & 1, kq + 1)
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int mi = 0; mi < 4; mi++)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[kq & 1][ni], fa[kq & 1][mi], acc[ni][mi], 0, 0, DG_NEG_A);
        }
#undef F64N_READ
    }
    if (product_counter && lane == 0 && T)
        atomicAdd(product_counter, (unsigned long long)(8 * T));

    double __attribute__((address_space(1))) *C = (double __attribute__((address_space(1))) *)reinterpret_cast<double *>(G.cdense);
    const unsigned c_voff = ((unsigned)l4 * (unsigned)nb + (unsigned)l15) * 8u;
#define F64N_C(ni_, mi_, r_)                                                                         \
    (*(double __attribute__((address_space(1))) *)(dg_scalar_base((const char __attribute__((address_space(1))) *)C +                    \
                                                                  ((size_t)(N0 + wn + (ni_) * 16 + 4 * (r_)) * nb + M0 + wm) * 8) + \
                                                   dg_lane_offset(c_voff) + (mi_) * 128))
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
    {
        double old[4][4];
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                old[mi][r] = F64N_C(ni, mi, r);
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                F64N_C(ni, mi, r) = old[mi][r] + acc[ni][mi][r];
    }
#undef F64N_C
}
