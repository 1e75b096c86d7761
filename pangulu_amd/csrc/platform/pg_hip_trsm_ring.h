// pg_hip_trsm_ring.h -- dense TSTRF / GESSM with the factor tiles requested AHEAD through a ring in LDS (round 5; included by
// pg_hip_platform.hip behind pg_hip_trsm_dense.h; R64 mirrors, nb = 128 or 256).  Replaces tstrf_cuda / gessm_cuda of the
// reference's GPU path (...0201000.cu:321-465); the arithmetic is trsm_dense_direct_f64_kernel's (pg_hip_trsm_dense.h): one
// wavefront solves one 16-wide strip panel by panel, solution tiles in registers, the 16 x 16 diagonal tiles inverted by the
// factorisation, products with structurally empty factor tiles skipped.
//
// Why another one.  A lone 256 x 256 solve takes 68 us (TSTRF) in the direct kernel where its matrix-core work is 15, and a batched
// launch runs at the same per-workgroup pace (two workgroups per CU: nothing hides a workgroup's waits).  The ISA says why: every
// factor tile is requested right in front of its own four products (all waits are vmcnt(3..0)).  The requests sit behind
// wavefront-uniform branches -- live-tile tests, skipped panels -- and the compiler counts a request under a condition as not issued
// when it places the waits; with 245 registers in use there is also no room for more than two tiles in flight.
//
// Measured (tools/microbench/bench_trsm.hip, results bit-identical to the direct kernel; profiles/r05a{s,t}_trsm_ring_*): TSTRF 1 / 64 /
// 1 024 blocks of 256 x 256: 46.8 us / 0.90 / 0.63 us per block against 69.0 / 1.29 / 1.12; GESSM 56.0 / 1.00 / 0.69 against 55.0 / 1.28 / 0.74.
//
// Here the factor tiles of a strip travel HBM/L2 -> LDS by `global_load_lds_dwordx4` (no registers), up to TR_RING - 1 tiles ahead
// of the matrix cores, in exactly the order the solve will use them (a scalar state machine walks the live tiles of the live
// panels, diagonal tile last); the consumer waits with a COUNTED vmcnt for the oldest tile only.  Every wavefront has a ring of its
// own (8 slots of 2 KiB): no barriers, a wavefront without live tiles leaves at once, as in the direct kernel.
//   slot layout = the order the DMA writes it: instruction h (0, 1), lane l, 2 doubles
//     TSTRF  A'[c][k] = U(16q + k, 16p + c):  lane l of instruction h fetches c = l & 15, k = 2 ((l >> 4) + 4 h) + {0, 1}
//            operand of k-quarter kq for lane (l15, l4):  slot[64 kq + 32 (l4 >> 1) + 2 l15 + (l4 & 1)]
//     GESSM  A [r][k] = L(16p + r, 16q + k):  lane l of instruction h fetches k = (l >> 3) + 8 h, r = 2 (l & 7) + {0, 1}
//            (eight lanes = 128 contiguous bytes of one column)
//            operand of k-quarter kq for lane (l15, l4):  slot[64 kq + 16 l4 + l15]   (the slot is the tile, column-major)
#pragma once

#define TR_RING 8
#ifndef TR_GESSM_THROUGH_RING
#define TR_GESSM_THROUGH_RING 1 // 0: GESSM tasks on the direct body (the first version of this kernel was slower than it on GESSM: 69 against 55 us alone)
#endif
#define TR_SLOT_DOUBLES 256

__host__ __device__ inline size_t tr_lds_bytes(int nb) { return (size_t)4 * TR_RING * TR_SLOT_DOUBLES * sizeof(double) + (size_t)4 * (nb / 16) * sizeof(unsigned); }

// wait until at most `newer` tiles (two DMA instructions each) requested after the one wanted are still on their way
__device__ __forceinline__ void tr_wait_tiles(int newer)
{
    switch (newer)
    {
    case 0:
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        break;
    case 1:
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        break;
    case 2:
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        break;
    case 3:
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        break;
    case 4:
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        break;
    case 5:
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        break;
    case 6:
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        break;
    default:
        asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        break;
    }
}

template <int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void trsm_dense_ring_f64_kernel(const TrsmDenseTaskD *__restrict__ tasks, const u32 *__restrict__ work)
{
    constexpr int nb = NP * 16;
    constexpr int slabs = NP * 16 / 64;
    extern __shared__ __align__(16) unsigned char tr_smem[];
    const u32 item = work[logical_block_id((unsigned)slabs)];
    const TrsmDenseTaskD T = tasks[item >> 2];
    const int slab = (int)(item & 3u);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int o0 = slab * 64 + wave * 16; // this wavefront's 16 rows (TSTRF) / columns (GESSM)
    double *ring = reinterpret_cast<double *>(tr_smem) + (size_t)wave * TR_RING * TR_SLOT_DOUBLES;
    unsigned *s_live = reinterpret_cast<unsigned *>(reinterpret_cast<double *>(tr_smem) + (size_t)4 * TR_RING * TR_SLOT_DOUBLES) + wave * NP;
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
    if (!tstrf && !TR_GESSM_THROUGH_RING)
    {
        // (kept for measurements: with the requests of a GESSM tile laid along its columns and the operands of the next tile read
        //  from LDS behind the current products the ring is the faster kernel for GESSM too: 0.69 against 0.74 us per task of 1 024)
        trsm_dense_direct_body<NP, false>(T, slab, wave, lane);
        return;
    }
    // live products of panel p: solution tile q live AND factor tile (q, p) live (as the direct kernel); plus bit p: the diagonal tile
    const unsigned short *fmap = mirror_map(LU, nb);
    const bool use_fmap = T.lu_map != 0;
    unsigned fm[NP];
#pragma unroll
    for (int c = 0; c < NP; c++)
        fm[c] = use_fmap ? (unsigned)__builtin_amdgcn_readfirstlane((int)fmap[c]) : 0xFFFFu;
    unsigned lqv[NP];
#pragma unroll
    for (int p = 0; p < NP; p++)
    {
        unsigned fl = 0;
        if (tstrf)
            fl = fm[p];
        else
        {
#pragma unroll
            for (int q = 0; q < NP; q++)
                fl |= ((fm[q] >> p) & 1u) << q;
        }
        lqv[p] = (my_lv & fl & ((1u << p) - 1u)) | (1u << p);
        if (lane == 0)
            s_live[p] = lqv[p];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (the list is read back by this wavefront only)

    typedef const char __attribute__((address_space(1))) *gbytes;
    typedef double __attribute__((address_space(3))) *lptr;
    // ---- the requester: tiles in the order of use, TR_RING - 1 ahead -------------------------------------------------------------
    const unsigned src_lane = (unsigned)(tstrf ? (lane & 15) * nb + 2 * (lane >> 4) : (lane >> 3) * nb + 2 * (lane & 7)) * 8u;
    const unsigned src_step = tstrf ? 64u : (unsigned)(8 * nb) * 8u; // second instruction of a tile
    int head = 0, tail = 0; // tiles requested / consumed
    int rp = NP;            // panel of the next request
    unsigned rmask = 0;     // its tiles still to request (bit q < rp: factor tile (q, rp); bit rp: the diagonal tile)
    {
        rp = __builtin_ctz(my_lv);
        rmask = (unsigned)__builtin_amdgcn_readfirstlane((int)s_live[rp]);
    }
    auto request_one = [&]()
    {
        if (rp >= NP)
            return;
        const int q = __builtin_ctz(rmask);
        rmask &= rmask - 1u;
        const gbytes ab = (gbytes)LU + (size_t)(tstrf ? (16 * rp) * nb + 16 * q : (16 * q) * nb + 16 * rp) * 8;
        double *dst = ring + (size_t)(head & (TR_RING - 1)) * TR_SLOT_DOUBLES;
#pragma unroll
        for (int h = 0; h < 2; h++)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(dg_scalar_base(ab + (size_t)h * src_step) + dg_lane_offset(src_lane)),
                                             (lptr)(dst + h * 128), 16, 0, 0);
        head++;
        if (!rmask)
        {
            const unsigned rest = my_lv & ~((2u << rp) - 1u);
            if (!rest)
                rp = NP;
            else
            {
                rp = __builtin_ctz(rest);
                rmask = (unsigned)__builtin_amdgcn_readfirstlane((int)s_live[rp]);
            }
        }
    };
    for (int i = 0; i < TR_RING - 1; i++)
        request_one();

    // ---- the strip's solution tiles (TSTRF: X^T tiles, GESSM: X tiles), as the direct kernel -------------------------------------------
    const unsigned x_voff = (unsigned)(tstrf ? l4 * nb + l15 : l15 * nb + l4) * 8u;
#define TR_X(p_, g_)                                                                                                \
    (*(double __attribute__((address_space(1))) *)(dg_scalar_base((gbytes)Bm + (tstrf ? (size_t)(16 * (p_) + 4 * (g_)) * nb + o0     \
                                                                                         : (size_t)o0 * nb + 16 * (p_) + 4 * (g_)) * 8) + \
                                                   dg_lane_offset(x_voff)))
    v4f64 xs[NP];
#pragma unroll
    for (int p = 0; p < NP; p++)
#pragma unroll
        for (int g = 0; g < 4; g++)
            xs[p][g] = !((my_lv >> p) & 1u) ? 0.0 : TR_X(p, g);
    // (a wait the compiler sees: the solution tiles and the first requests arrive together, and the counted waits below then count
    //  DMA instructions only)
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)

    // ---- the consumer ---------------------------------------------------------------------------------------------------------------
    const int rd_lane = tstrf ? 32 * (l4 >> 1) + 2 * l15 + (l4 & 1) : 16 * l4 + l15;
    // (the operands of the NEXT tile are read from its slot while the matrix cores work on the current one: a tile's fixed cost --
    //  the wait, four LDS reads and their latency, the requester's scalar code -- is what bounds a lone solve, not the trip to L2)
    double a_pf[4];
    auto prefetch = [&]()
    {
        if (head == tail)
            return; // (nothing left)
        if (rp < NP)
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); // (steady state: TR_RING - 1 tiles requested, the oldest wanted)
        else
            tr_wait_tiles(head - tail - 1);
        const lptr slot = (lptr)ring + ((tail & (TR_RING - 1)) * TR_SLOT_DOUBLES + rd_lane);
#pragma unroll
        for (int kq = 0; kq < 4; kq++)
            a_pf[kq] = slot[64 * kq];
    };
    prefetch();
    auto consume = [&](double(&a)[4])
    {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (the slot is free for the next request)
#pragma unroll
        for (int kq = 0; kq < 4; kq++)
            a[kq] = a_pf[kq];
        tail++;
        request_one();
        prefetch();
    };
#pragma unroll
    for (int p = 0; p < NP; p++)
    {
        if (!((my_lv >> p) & 1u))
            continue; // (wavefront-uniform; the requester skips the panel as well)
        const unsigned lq = lqv[p];
        v4f64 part[4];
        part[0] = xs[p];
        part[1] = part[2] = part[3] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < p; q++)
        {
            if (!((lq >> q) & 1u))
                continue;
            double a[4];
            consume(a);
#pragma unroll
            for (int kq = 0; kq < 4; kq++)
                part[kq] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kq], xs[q][kq], part[kq], 0, 0, 1); // (NEG field: part -= a x)
        }
        double ad[4];
        consume(ad);
        v4f64 acc = (part[0] + part[1]) + (part[2] + part[3]);
        // multiply by the inverted diagonal tile (upper part: inv(U_pp); strictly lower part: inv(L_pp), unit diagonal)
        v4f64 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kq = 0; kq < 4; kq++)
        {
            const int k = kq * 4 + l4;
            double a;
            if (tstrf)
                a = (k <= l15) ? ad[kq] : 0.0;
            else
                a = (l15 > k) ? ad[kq] : ((l15 == k) ? 1.0 : 0.0);
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[kq], x, 0, 0, 0);
        }
        xs[p] = x;
    }
#pragma unroll
    for (int p = 0; p < NP; p++)
    {
        if (!((my_lv >> p) & 1u))
            continue;
#pragma unroll
        for (int g = 0; g < 4; g++)
            TR_X(p, g) = xs[p][g];
    }
#undef TR_X
}
