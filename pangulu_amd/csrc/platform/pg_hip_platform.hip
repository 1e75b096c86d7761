// pg_hip_platform.hip -- the MI355X (gfx950 / CDNA4) back-end behind the platform C-ABI (include/pangulu_platform.h).
//
// Replaces the reference's CUDA back-end (src/platforms/02_NONSHAREDMEM/01_GPU/000_CUDA/pangulu_platform_0201000.cu)
// with kernels designed for 64-wide wavefronts, 160 KB of LDS per CU and the f64 matrix cores:
//
//   SSSSM  sparse : ONE launch per batch (the reference launches one kernel per task, ...0201000.cu:856-863, from a
//                   serial host loop, :875-898).  Tasks are grouped by destination block; one wavefront owns one
//                   destination column for ALL updates of the group: the column is scattered once into a dense LDS
//                   vector, every op2 entry of that column streams the matching op1 column through it with
//                   coalesced CSC walks, and the column is gathered back once.  No atomics at all (the reference
//                   uses shared + global atomicAdd, :467-545), results are deterministic.
//   SSSSM  dense  : destination, op1 and op2 completely full -> the value arrays ARE column-major nb x nb
//                   matrices (reference rule ...0201000.cu:827); C -= sum_t A_t * B_t runs on
//                   v_mfma_f64_16x16x4_f64 with the accumulators kept in registers across all tasks of a group.
//   TSTRF / GESSM : one wavefront per row / column of the block in a dense LDS vector (batched over blocks).
//   GETRF         : one 1024-thread workgroup per diagonal block, right-looking elimination on a dense scratch
//                   image that is only ever touched on the block's (symbolically closed) pattern.
//
// CPU semantics being matched: ...01_SHAREDMEM/00_CPU/000_CPU/pangulu_platform_0100000.c:57-431 (incl. the
// PANGULU_TOL pivot clamp of :79-84,152-157, which the reference's GPU path lacks).
#include <hip/hip_runtime.h>

#include <sched.h>
#include <cctype>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <thread>
#include <mutex>
#include <functional>
#include <tuple>
#include <type_traits>
#include <unordered_map>
#include <unordered_set>
#include <unordered_set>
#include <vector>

#include "../../../include/pangulu_platform.h"

typedef calculate_type val_t;
typedef calculate_real_type real_t;
typedef pangulu_storage_slot_t slot_t;
typedef pangulu_task_t task_t;
typedef unsigned int u32;
typedef unsigned short u16;
typedef unsigned long long u64;

#define HIP_CHECK(expr)                                                                                              \
    do                                                                                                               \
    {                                                                                                                \
        hipError_t e_ = (expr);                                                                                      \
        if (e_ != hipSuccess)                                                                                        \
        {                                                                                                            \
            fprintf(stderr, "[PanguLU-AMD ERROR] HIP error at %s:%d %s (code=%d)\n", __FILE__, __LINE__,             \
                    hipGetErrorString(e_), (int)e_);                                                                 \
            exit(EXIT_FAILURE);                                                                                      \
        }                                                                                                            \
    } while (0)

// weak: the reference host defines these (src/pangulu.c:7-9)
extern "C"
{
    __attribute__((weak)) int pangulu_gpu_kernel_warp_per_block = 4;
    __attribute__((weak)) int pangulu_gpu_data_move_warp_per_block = 4;
    __attribute__((weak)) int pangulu_gpu_shared_mem_size = 0;
}

// -----------------------------------------------------------------------------------------------------------------
// value arithmetic (HIP has no _Complex; complex types are (re, im) pairs like the reference hand-expands)
// -----------------------------------------------------------------------------------------------------------------
#ifdef PANGULU_COMPLEX
__host__ __device__ inline val_t v_make(real_t r) { return val_t{r, (real_t)0}; }
__host__ __device__ inline val_t v_mul(val_t a, val_t b) { return val_t{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__host__ __device__ inline val_t v_sub(val_t a, val_t b) { return val_t{a.re - b.re, a.im - b.im}; }
// a - b*c
__host__ __device__ inline val_t v_submul(val_t a, val_t b, val_t c)
{
    return val_t{a.re - (b.re * c.re - b.im * c.im), a.im - (b.re * c.im + b.im * c.re)};
}
__host__ __device__ inline val_t v_div(val_t a, val_t b)
{
    real_t d = b.re * b.re + b.im * b.im;
    return val_t{(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}
__host__ __device__ inline real_t v_realpart(val_t a) { return a.re; }
#else
__host__ __device__ inline val_t v_make(real_t r) { return r; }
__host__ __device__ inline val_t v_mul(val_t a, val_t b) { return a * b; }
__host__ __device__ inline val_t v_sub(val_t a, val_t b) { return a - b; }
__host__ __device__ inline val_t v_submul(val_t a, val_t b, val_t c) { return a - b * c; } // contracts to one FMA
__host__ __device__ inline val_t v_div(val_t a, val_t b) { return a / b; }
__host__ __device__ inline real_t v_realpart(val_t a) { return a; }
#endif

// LDS accumulator -= v with hardware floating-point atomics
__device__ inline void lds_atomic_sub(val_t *dst, val_t v)
{
#ifdef PANGULU_COMPLEX
    atomicAdd(&dst->re, -v.re);
    atomicAdd(&dst->im, -v.im);
#else
    atomicAdd(dst, -v);
#endif
}

// dst += v with hardware floating-point atomics (skips exact zeros: most of a dense column of a sparse update)
__device__ inline void v_atomic_add(val_t *dst, val_t v)
{
#ifdef PANGULU_COMPLEX
    if (v.re != 0)
        atomicAdd(&dst->re, v.re);
    if (v.im != 0)
        atomicAdd(&dst->im, v.im);
#else
    if (v != 0)
        atomicAdd(dst, v);
#endif
}

// the CPU path's pivot clamp (...0100000.c:79-84): |real part| < 1e-16 -> +1e-16
__device__ inline val_t clamp_pivot(val_t p)
{
    real_t r = v_realpart(p);
    if ((r < 0 ? -r : r) < (real_t)PANGULU_TOL)
        return v_make((real_t)PANGULU_TOL);
    return p;
}

// Dense-mode updates (mirrors + the f64 MFMA update kernel): R64, and CR64 as two real planes per mirror (pg_hip_dense.h).
// Dense-mode PANELS (blocked GETRF, dense TSTRF/GESSM on LU images) exist for the real types.
// R32 / CR32 (round 3): the mirrors and LU images of single-precision blocks are DOUBLE -- densify widens, sparsify rounds --
// and every dense kernel is the f64 one: the arithmetic between two rounding points is at least the reference's (cuBLAS
// sgemm / cgemm on densified blocks, ...0201000.cu:778-816), one code path serves all four types, and the f64 matrix
// cores of gfx950 run at half the f32 rate, not at a fraction of it.
#define PG_DENSE_UPDATES 1
#if defined(CALCULATE_TYPE_R64) || defined(CALCULATE_TYPE_R32)
#define PG_DENSE_PANELS 1
#endif
#if defined(CALCULATE_TYPE_CR64) || defined(CALCULATE_TYPE_CR32)
#define PG_COMPLEX_PANELS 1 // GETRF / TSTRF / GESSM of dense-mode blocks on the two-plane mirrors (pg_hip_panels_complex.h)
#define PG_PLANES 2
#else
#define PG_PLANES 1
#endif

// -----------------------------------------------------------------------------------------------------------------
// device-side descriptors
// -----------------------------------------------------------------------------------------------------------------
struct BlkView // one compressed view of a block: pointer array over the major dimension, minor indices, values
{
    const u32 *ptr;
    const u16 *idx;
    val_t *val;
};

struct SsssmTaskD
{
    BlkView a; // op1 (L block), CSC
    BlkView b; // op2 (U block), CSC
    // MFMA kernel only.  CR64: a complex update is four real products on the planes of the mirrors; `sign` multiplies the A
    // operand (the A_im B_im product ADDS to the real plane), `count` marks the one of the four whose structural flops count
    double sign;
    u32 count;
    // has_map: amap / bmap_t hold the occupancy of the operands (from the host's summaries, BlockState::occ_map):
    //   amap[s]   bit r: A has pattern entries in row slab r of K-slab (column slab) s
    //   bmap_t[s] bit c: B has pattern entries in column slab c of K-slab (row slab) s
    // otherwise (blocks received from another rank) the kernel reads the maps behind the mirrors
    u32 has_map;
    unsigned short amap[16], bmap_t[16];
};
static_assert(sizeof(SsssmTaskD) == 128, "two task descriptors per 256 bytes");

// one workgroup of the MFMA update launch: a 128 x 128 tile of one destination and the queue of updates that reach it
struct SsssmWorkD
{
    val_t *cdense;          // the destination's mirror (CR64: one plane of it)
    u32 task_begin, task_end;
    u32 atomic, slab_mask;  // as in SsssmGroupD
    u32 tile, pad_;
};

struct SsssmGroupD
{
    BlkView c;        // destination CSC (diagonal destination: its strictly-lower half)
    // diagonal destination only: column view of the upper half (built once per diagonal block, see DiagAux)
    const u32 *ucp;
    const u16 *uri;
    const u32 *uvi;
    val_t *uval;
    // dense-mode destination: its nb x nb column-major mirror (see pg_hip_dense.h); the sparse views above are unused
    val_t *cdense;
    u32 task_begin, task_end;
    // a destination with many queued updates is cut into several groups that run concurrently: each then starts
    // from zero and ADDS its partial sum to the destination with floating-point atomics (otherwise the longest
    // queue sets the duration of the whole launch)
    u32 atomic;
    // MFMA kernel, nb <= 256: bit s set = this group works on K-slab s (16 wide) of every task; 0 = all slabs.  A launch
    // with a handful of updates (the diagonal block's update near the root of the tree, on the critical path of
    // every level) is cut four ways along K so that 16 CUs instead of 4 share a 256 x 256 x 256 product.
    u32 slab_mask;
    // host side only: which of the destination's 128 x 128 tiles some update of this group can reach (bit = tile index);
    // the launch leaves the others out (pangulu_platform_0201001_prepare_blocks)
    u32 live_tiles;
    u32 pad_;
};

struct TrsmTaskD
{
    // vectors of the block being solved: TSTRF walks rows (CSR view + map into the CSC values), GESSM columns
    const u32 *vptr;
    const u16 *vidx;
    const u32 *vmap; // nullptr for GESSM
    val_t *bval;
    // triangular factor: TSTRF: upper half, CSR, diagonal first in each row; GESSM: strictly-lower CSC, unit diagonal
    const u32 *tptr;
    const u16 *tidx;
    const val_t *tval;
    u32 is_tstrf;
    u32 pad_;
};

struct GetrfTaskD
{
    const u32 *lcp;
    const u16 *lri;
    val_t *lval;
    const u32 *urp;
    const u16 *uci;
    val_t *uval;
    val_t *dense; // nb*nb scratch, only touched on the pattern (the blocked kernels: a DOUBLE image, whatever val_t is)
    u32 preloaded; // blocked kernel: `dense` already holds the block (a dense-mode mirror): skip zero + scatter
    u32 invert_tiles; // blocked kernel: the image will serve the dense solves -- replace its 16 x 16 diagonal tiles by their
                      // inverses before leaving (pg_hip_trsm_dense.h), after the factors have been gathered / the tiles saved
    u32 defer_gather; // blocked kernel on a mirror: leave the factors in the dense image, save the 16 x 16 diagonal tiles
                      // behind the mirror (they are about to be inverted in place) -- a sparsify job on the records
                      // stream brings the sparse record up to date off the critical path
};

__device__ inline u32 ptr0(const u32 *p, int i) { return i == 0 ? 0u : p[i]; }

// Workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  The `per_unit` workgroups that read the same
// operands (the column runs of one update queue, the tiles of one destination, the strips of one solve) are mapped
// to the SAME XCD so that they share its L2; units themselves stay dealt round-robin (unit u -> XCD u % 8), which keeps
// the longest-first launch order balanced over the XCDs.  Returns unit * per_unit + index inside the unit.
__constant__ int c_xcd_swizzle = 1;
__device__ inline unsigned logical_block_id(unsigned per_unit)
{
    const unsigned n = gridDim.x, b = blockIdx.x;
    const unsigned round = 8u * per_unit, full = (n / round) * round; // workgroups in complete rounds of 8 units
    if (!c_xcd_swizzle || b >= full)
        return b;
    const unsigned x = b & 7, idx = b >> 3;
    return (x + 8u * (idx / per_unit)) * per_unit + idx % per_unit;
}

__device__ inline void wave_lds_fence()
{
    // LDS operations of one wavefront execute in order; this only stops the compiler from reordering them
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ inline unsigned long long wave_sum(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, 64);
    return v;
}

// a dense-mode destination as the sparse update kernel sees it: R64 one nb x nb image, CR64 two real planes (pg_hip_dense.h)
__host__ __device__ inline size_t cdense_plane_stride(int nb) { return (size_t)nb * nb + 64 / sizeof(double) + (size_t)16 * nb; }
__device__ inline val_t cdense_get(const val_t *cd, size_t at, int nb)
{
#if PG_PLANES > 1
    const double *p = reinterpret_cast<const double *>(cd);
    return val_t{(real_t)p[at], (real_t)p[at + cdense_plane_stride(nb)]};
#else
    (void)nb;
    return (val_t) reinterpret_cast<const double *>(cd)[at]; // (the mirror of a single-precision block is double as well)
#endif
}
__device__ inline void cdense_put(val_t *cd, size_t at, int nb, val_t v)
{
#if PG_PLANES > 1
    double *p = reinterpret_cast<double *>(cd);
    p[at] = v.re;
    p[at + cdense_plane_stride(nb)] = v.im;
#else
    (void)nb;
    reinterpret_cast<double *>(cd)[at] = v;
#endif
}
__device__ inline void cdense_atomic_add(val_t *cd, size_t at, int nb, val_t v)
{
#if PG_PLANES > 1
    double *p = reinterpret_cast<double *>(cd);
    if (v.re != 0)
        atomicAdd(&p[at], (double)v.re);
    if (v.im != 0)
        atomicAdd(&p[at + cdense_plane_stride(nb)], (double)v.im);
#else
    (void)nb;
    if (v != 0)
        atomicAdd(&reinterpret_cast<double *>(cd)[at], (double)v);
#endif
}

// -----------------------------------------------------------------------------------------------------------------
// SSSSM, sparse.  grid = groups * ceil(nb / WAVES) workgroups of WAVES wavefronts; wave w of block (g, cb) owns
// destination column cb*WAVES + w of group g.
// -----------------------------------------------------------------------------------------------------------------
#define SSSSM_WAVES 4
#ifndef SSSSM_SUBW
#define SSSSM_SUBW 8 // lanes per op2 entry in the non-strict kernel (64 / SSSSM_SUBW entries in flight per wavefront)
#endif

// STRICT = true : one op2 entry at a time over the whole wavefront, every update a fused multiply-add applied in
//                  ascending k: the order the FMA oracle restates, results reproducible bit for bit.
// STRICT = false: four op2 entries at a time, one per quarter-wavefront (op1 columns of sparse blocks are short: a
//                  full wavefront per column leaves most lanes idle and serialises on L2 latency); the four columns
//                  may hit the same accumulator row in one instruction, so the LDS accumulator takes hardware
//                  floating-point atomics (ds_add_f64 / ds_add_f32).  Products are rounded before the add.
template <bool STRICT>
__global__ __launch_bounds__(SSSSM_WAVES * 64) void ssssm_sparse_kernel(const SsssmGroupD *__restrict__ groups,
                                                                         const SsssmTaskD *__restrict__ tasks, int nb, int cpw,
                                                                         unsigned long long *flop_counter)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    val_t *smem = reinterpret_cast<val_t *>(smem_raw);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // a wavefront owns `cpw` adjacent destination columns, one after the other (big batches use cpw > 1: launching a
    // workgroup per 4 columns costs more than the columns themselves, most of which no update touches)
    const int colblocks = (nb + SSSSM_WAVES * cpw - 1) / (SSSSM_WAVES * cpw);
    const unsigned bid = logical_block_id((unsigned)colblocks);
    const int g = bid / colblocks;
    const int jbase = ((bid % colblocks) * SSSSM_WAVES + wave) * cpw;
    val_t *acc = smem + (size_t)wave * nb;
    const SsssmGroupD G = groups[g];
    unsigned long long fmas = 0;
    for (int j = jbase; j < jbase + cpw && j < nb; j++)
    {
    // does any update of the group touch column j at all?  (cheap uniform scan; most columns of sparse blocks don't)
    bool any = false;
    for (u32 t = G.task_begin; t < G.task_end && !any; t++)
    {
        const u32 *bp = tasks[t].b.ptr;
        any = bp[j + 1] > ptr0(bp, j);
    }
    if (!any)
        continue;

    u32 c0 = 0, c1 = 0, u0 = 0, u1 = 0;
    const bool atomic = G.atomic != 0;
    if (G.cdense)
    {
        // dense-mode destination: the column is a contiguous run of its mirror
        for (int r = lane; r < nb; r += 64)
            acc[r] = atomic ? v_make(0) : cdense_get(G.cdense, (size_t)j * nb + r, nb);
    }
    else
    {
        c0 = ptr0(G.c.ptr, j);
        c1 = G.c.ptr[j + 1];
        for (u32 p = c0 + lane; p < c1; p += 64)
            acc[G.c.idx[p]] = atomic ? v_make(0) : G.c.val[p];
        if (G.ucp)
        {
            u0 = G.ucp[j];
            u1 = G.ucp[j + 1];
            for (u32 p = u0 + lane; p < u1; p += 64)
                acc[G.uri[p]] = atomic ? v_make(0) : G.uval[G.uvi[p]];
        }
    }
    wave_lds_fence();

    for (u32 t = G.task_begin; t < G.task_end; t++)
    {
        const BlkView A = tasks[t].a, B = tasks[t].b;
        const u32 b0 = ptr0(B.ptr, j), b1 = B.ptr[j + 1];
        for (u32 qb = b0; qb < b1; qb += 64)
        {
            // one coalesced read of up to 64 entries of op2's column, then broadcast them one by one
            const u32 cnt = min(64u, b1 - qb);
            u32 my_k = 0, my_a0 = 0, my_a1 = 0;
            val_t my_b = v_make(0);
            if ((u32)lane < cnt)
            {
                my_k = B.idx[qb + lane];
                my_b = B.val[qb + lane];
                my_a0 = ptr0(A.ptr, (int)my_k);
                my_a1 = A.ptr[my_k + 1];
            }
            if (!STRICT)
            {
                const int sub = lane / SSSSM_SUBW, sl = lane % SSSSM_SUBW;
                for (u32 i = 0; i < cnt; i += 64 / SSSSM_SUBW)
                {
                    const int src = (int)min(i + (u32)sub, cnt - 1);
                    // (all lanes take part in every shuffle: a lane may be the source of another quarter's read)
                    const u32 a0 = __shfl(my_a0, src, 64);
                    const u32 a1s = __shfl(my_a1, src, 64);
                    const u32 a1 = (i + (u32)sub < cnt) ? a1s : a0;
#ifdef PANGULU_COMPLEX
                    val_t bv;
                    bv.re = __shfl(my_b.re, src, 64);
                    bv.im = __shfl(my_b.im, src, 64);
#else
                    const val_t bv = __shfl(my_b, src, 64);
#endif
                    for (u32 r = a0 + (u32)sl; r < a1; r += SSSSM_SUBW)
                    {
                        const val_t prod = v_mul(A.val[r], bv);
                        lds_atomic_sub(&acc[A.idx[r]], prod);
                        fmas++;
                    }
                }
                wave_lds_fence();
                continue;
            }
            // four op2 entries per pass: the first 64 elements of their four op1 columns are requested together
            // (the walk is latency-bound: one column in flight per wavefront leaves the memory pipe idle), then
            // applied one after the other -- the order of updates per destination entry stays ascending in k
            for (u32 i = 0; i < cnt; i += 4)
            {
                u32 a0[4], a1[4], row[4];
                val_t bv[4], av[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    const int src = (int)min(i + (u32)u, cnt - 1);
                    a0[u] = __shfl(my_a0, src, 64);
                    a1[u] = (i + (u32)u < cnt) ? __shfl(my_a1, src, 64) : a0[u];
#ifdef PANGULU_COMPLEX
                    bv[u].re = __shfl(my_b.re, src, 64);
                    bv[u].im = __shfl(my_b.im, src, 64);
#else
                    bv[u] = __shfl(my_b, src, 64);
#endif
                    const u32 r = a0[u] + (u32)lane;
                    ok[u] = r < a1[u];
                    row[u] = 0;
                    av[u] = v_make(0);
                    if (ok[u])
                    {
                        row[u] = A.idx[r];
                        av[u] = A.val[r];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    if (ok[u])
                    {
                        acc[row[u]] = v_submul(acc[row[u]], av[u], bv[u]);
                        fmas++;
                    }
                    for (u32 r = a0[u] + 64 + lane; r < a1[u]; r += 64)
                    {
                        const u32 rr = A.idx[r];
                        acc[rr] = v_submul(acc[rr], A.val[r], bv[u]);
                        fmas++;
                    }
                    wave_lds_fence();
                }
            }
        }
    }

    if (atomic)
    {
        if (G.cdense)
        {
            for (int r = lane; r < nb; r += 64)
                cdense_atomic_add(G.cdense, (size_t)j * nb + r, nb, acc[r]);
        }
        else
        {
            for (u32 p = c0 + lane; p < c1; p += 64)
                v_atomic_add(&G.c.val[p], acc[G.c.idx[p]]);
            if (G.ucp)
                for (u32 p = u0 + lane; p < u1; p += 64)
                    v_atomic_add(&G.uval[G.uvi[p]], acc[G.uri[p]]);
        }
    }
    else if (G.cdense)
    {
        for (int r = lane; r < nb; r += 64)
            cdense_put(G.cdense, (size_t)j * nb + r, nb, acc[r]);
    }
    else
    {
        for (u32 p = c0 + lane; p < c1; p += 64)
            G.c.val[p] = acc[G.c.idx[p]];
        if (G.ucp)
        {
            for (u32 p = u0 + lane; p < u1; p += 64)
                G.uval[G.uvi[p]] = acc[G.uri[p]];
        }
    }
    wave_lds_fence(); // the accumulator is reused for the next column
    } // columns of this wavefront
    fmas = wave_sum(fmas);
    if (lane == 0 && fmas)
        atomicAdd(flop_counter, 2ull * fmas);
}

// -----------------------------------------------------------------------------------------------------------------
// SSSSM, dense (R64) and the dense-mode mirrors: pg_hip_dense.h
// -----------------------------------------------------------------------------------------------------------------
#if defined(PG_DENSE_UPDATES)
typedef double v4f64 __attribute__((ext_vector_type(4)));
#include "pg_hip_dense.h"
#include "pg_hip_front.h"
#endif
#if defined(PG_DENSE_PANELS) || defined(PG_COMPLEX_PANELS)
#include "pg_hip_trsm_dense.h"
#endif
#if defined(PG_COMPLEX_PANELS)
#include "pg_hip_panels_complex.h"
#endif

// -----------------------------------------------------------------------------------------------------------------
// TSTRF / GESSM.  grid = tasks * ceil(nb / 4); wave w owns one row (TSTRF) or column (GESSM) of the block.
//   TSTRF (...0100000.c:137-175): for each structural entry c of the row, ascending: x_c /= U(c,c), then
//                                 x_{c'} -= x_c * U(c,c') for the tail of U's row c.
//   GESSM (...0100000.c:178-209): for each entry r of the column, ascending: x_{r'} -= x_r * L(r',r).
// The pattern is closed under these updates, so every touched x entry belongs to the vector's own pattern.
// -----------------------------------------------------------------------------------------------------------------
#define TRSM_WAVES 4

__global__ __launch_bounds__(TRSM_WAVES * 64) void trsm_sparse_kernel(const TrsmTaskD *__restrict__ tasks, int nb,
                                                                       unsigned long long *flop_tstrf,
                                                                       unsigned long long *flop_gessm)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    val_t *smem = reinterpret_cast<val_t *>(smem_raw);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int vblocks = (nb + TRSM_WAVES - 1) / TRSM_WAVES;
    const unsigned bid = logical_block_id((unsigned)vblocks);
    const int t = bid / vblocks;
    const int v = (bid % vblocks) * TRSM_WAVES + wave;
    if (v >= nb)
        return;
    const TrsmTaskD T = tasks[t];
    const u32 s = ptr0(T.vptr, v), e = T.vptr[v + 1];
    if (s == e)
        return;
    val_t *x = smem + (size_t)wave * nb;
    for (u32 p = s + lane; p < e; p += 64)
        x[T.vidx[p]] = T.bval[T.vmap ? T.vmap[p] : p];
    wave_lds_fence();

    unsigned long long ops = 0;
    for (u32 pb = s; pb < e; pb += 64)
    {
        const u32 cnt = min(64u, e - pb);
        u32 my_c = 0, my_t0 = 0, my_t1 = 0;
        if ((u32)lane < cnt)
        {
            my_c = T.vidx[pb + lane];
            my_t0 = T.tptr[my_c];
            my_t1 = T.tptr[my_c + 1];
        }
        // The steps of one vector are strictly sequential, so the only parallelism to be had is in the memory
        // pipe: while step i is applied, the pivot and the first 64 factor entries of step i+1 are already in flight.
        u32 n_c, n_t0, n_t1, n_k = 0;
        val_t n_d = v_make(1), n_v = v_make(0);
        bool n_ok;
#define TRSM_FETCH(i_)                                                   \
    {                                                                    \
        n_c = __shfl(my_c, (int)(i_), 64);                               \
        n_t0 = __shfl(my_t0, (int)(i_), 64);                             \
        n_t1 = __shfl(my_t1, (int)(i_), 64);                             \
        if (T.is_tstrf)                                                  \
        {                                                                \
            n_d = T.tval[n_t0];                                          \
            n_t0++;                                                      \
        }                                                                \
        const u32 r_ = n_t0 + (u32)lane;                                 \
        n_ok = r_ < n_t1;                                                \
        if (n_ok)                                                        \
        {                                                                \
            n_k = T.tidx[r_];                                            \
            n_v = T.tval[r_];                                            \
        }                                                                \
    }
        TRSM_FETCH(0)
        for (u32 i = 0; i < cnt; i++)
        {
            const u32 c = n_c, t0 = n_t0, t1 = n_t1, k0 = n_k;
            const val_t d = n_d, v0 = n_v;
            const bool ok = n_ok;
            if (i + 1 < cnt)
                TRSM_FETCH(i + 1)
            val_t xc = x[c];
            if (T.is_tstrf)
            {
                xc = v_div(xc, clamp_pivot(d));
                if (lane == 0)
                {
                    x[c] = xc;
                    ops += 1;
                }
            }
            if (ok)
            {
                x[k0] = v_submul(x[k0], xc, v0);
                ops += 2;
            }
            for (u32 r = t0 + 64 + lane; r < t1; r += 64)
            {
                const u32 k = T.tidx[r];
                x[k] = v_submul(x[k], xc, T.tval[r]);
                ops += 2;
            }
            wave_lds_fence();
        }
#undef TRSM_FETCH
    }
    for (u32 p = s + lane; p < e; p += 64)
        T.bval[T.vmap ? T.vmap[p] : p] = x[T.vidx[p]];
    ops = wave_sum(ops);
    if (lane == 0 && ops)
        atomicAdd(T.is_tstrf ? flop_tstrf : flop_gessm, ops);
}

// -----------------------------------------------------------------------------------------------------------------
// GETRF (...0100000.c:57-135): one workgroup per diagonal block.  The block is expanded into a column-major
// dense image that is read and written ONLY on the block's pattern (closed under elimination), so it needs no
// clearing; per pivot k: scale L(:,k), then the rank-1 update over L(:,k) x U(k,:) spread over the workgroup.
// Every entry receives its updates in ascending k, the same order as the CPU merges.
// -----------------------------------------------------------------------------------------------------------------
#define GETRF_THREADS 1024

__global__ __launch_bounds__(GETRF_THREADS) void getrf_kernel(const GetrfTaskD *__restrict__ tasks, int nb,
                                                              unsigned long long *flop_counter)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    val_t *sL = reinterpret_cast<val_t *>(smem_raw); // nb values of L(:,k)
    val_t *sU = sL + nb;                             // nb values of U(k,:)
    u16 *sLi = reinterpret_cast<u16 *>(sU + nb);     // their rows
    u16 *sUi = sLi + nb;                             // their columns
    const GetrfTaskD T = tasks[blockIdx.x];
    val_t *D = T.dense;
    const int tid = threadIdx.x;

    for (int c = tid; c < nb; c += GETRF_THREADS)
    {
        for (u32 p = T.lcp[c]; p < T.lcp[c + 1]; p++)
            D[(size_t)c * nb + T.lri[p]] = T.lval[p];
        for (u32 p = T.urp[c]; p < T.urp[c + 1]; p++) // here c is a row of the CSR half
            D[(size_t)T.uci[p] * nb + c] = T.uval[p];
    }
    __syncthreads();

    unsigned long long ops = 0;
    for (int k = 0; k < nb; k++)
    {
        const u32 u0 = T.urp[k], u1 = T.urp[k + 1];
        if (u0 == u1)
            continue; // uniform
        const u32 l0 = T.lcp[k], l1 = T.lcp[k + 1];
        const int nL = (int)(l1 - l0), nU = (int)(u1 - u0 - 1);
        if (nL == 0)
            continue; // nothing below the pivot: neither L nor the trailing block changes
        const val_t pivot = clamp_pivot(D[(size_t)k * nb + k]);
        for (int i = tid; i < nL; i += GETRF_THREADS)
        {
            const u32 r = T.lri[l0 + i];
            const val_t v = v_div(D[(size_t)k * nb + r], pivot);
            D[(size_t)k * nb + r] = v;
            sL[i] = v;
            sLi[i] = (u16)r;
            ops += 1;
        }
        for (int jx = tid; jx < nU; jx += GETRF_THREADS)
        {
            const u32 c = T.uci[u0 + 1 + jx];
            sU[jx] = D[(size_t)c * nb + k];
            sUi[jx] = (u16)c;
        }
        __syncthreads();
        // rank-1 update over L(:,k) x U(k,:): a wavefront takes (a power-of-two group of) columns of U(k,:) and
        // its lanes the rows of L(:,k); short L columns pack several U columns into one wavefront
        {
            const int wave = tid >> 6, lane = tid & 63;
            int lg = 0;
            while ((1 << lg) < nL && lg < 6)
                lg++;
            const int P = 1 << lg;           // lanes per column (>= min(nL, 64), power of two)
            const int cpw = 64 >> lg;        // columns per wavefront pass
            const int isub = lane & (P - 1), jsub = lane >> lg;
            for (int jx = wave * cpw + jsub; jx < nU; jx += (GETRF_THREADS / 64) * cpw)
            {
                const val_t u = sU[jx];
                const size_t colbase = (size_t)sUi[jx] * nb;
                for (int i = isub; i < nL; i += P)
                {
                    const size_t off = colbase + sLi[i];
                    D[off] = v_submul(D[off], sL[i], u);
                    ops += 2;
                }
            }
        }
        __syncthreads();
    }

    for (int c = tid; c < nb; c += GETRF_THREADS)
    {
        for (u32 p = T.lcp[c]; p < T.lcp[c + 1]; p++)
            T.lval[p] = D[(size_t)c * nb + T.lri[p]];
        for (u32 p = T.urp[c]; p < T.urp[c + 1]; p++)
            T.uval[p] = D[(size_t)T.uci[p] * nb + c];
    }
    ops = wave_sum(ops);
    if ((tid & 63) == 0 && ops)
        atomicAdd(flop_counter, ops);
}

// -----------------------------------------------------------------------------------------------------------------
// GETRF, blocked (R64, nb a multiple of 16 up to 256).  The kernel above pays two L2 round trips per pivot
// (~1.5 us x nb): it is latency-bound whatever the fill.  This one keeps the active 16-column panel and the matching
// 16-row strip of U in LDS, eliminates inside them (LDS latency only), and applies the panel's rank-16 update to the
// trailing block on the f64 matrix cores straight from those LDS images:
//   for each panel j0:  P = D[j0:, j0:j0+16] (LDS, column-major)   S = D[j0:j0+16, j0+16:] (LDS, row-major)
//       16 pivots: scale L(:,k); rank-1 update of the rest of P and of S          (wavefront per column / row)
//       write P and S back;  D[j0+16:, j0+16:] -= P_lower * S                      (v_mfma_f64_16x16x4_f64)
// The dense image is zero outside the pattern; structural zeros make exact no-ops, so the factors equal the sparse
// algorithm's on the pattern.  Every entry still receives its updates in ascending pivot order.
// The trailing product is formed transposed (A operand = -S^T, B operand = P^T) so that each accumulator register
// maps to 16 consecutive rows of one column of D: loads and stores of the trailing block are 128-byte segments.
// -----------------------------------------------------------------------------------------------------------------
#if defined(PG_DENSE_PANELS)
#define GETRF_PANEL 16
#define GETRF_BLOCKED_ROWS 256 // one row thread per row: nb <= 256

// index i with ptr[i] <= p < ptr[i+1] (ptr ascending, ptr[0] = 0, p < ptr[n])
__device__ inline int owner_of(const u32 *ptr, int n, u32 p)
{
    int lo = 0, hi = n; // invariant: ptr[lo] <= p < ptr[hi]
    while (hi - lo > 1)
    {
        const int mid = (lo + hi) >> 1;
        if (ptr[mid] <= p)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

#define GETRF_STAMP(slot)                                                  \
    if (dbg && tid == 0 && blockIdx.x == 0)                                \
    {                                                                      \
        unsigned long long now_ = __builtin_amdgcn_s_memtime();            \
        dbg[slot] += now_ - stamp_;                                        \
        stamp_ = now_;                                                     \
    }

// THREADS = 1024: sixteen wavefronts, the whole register file of the CU (fastest for a block on its own).
// THREADS = 512: eight wavefronts capped at 128 registers -- half of the CU stays free, so the update and densify
// workgroups of a look-ahead batch run on the same CUs beside a launch that has a diagonal block for every CU.
template <int GETRF_BLOCKED_THREADS>
__global__ __launch_bounds__(GETRF_BLOCKED_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void getrf_blocked_f64_kernel(const GetrfTaskD *__restrict__ tasks, int nb,
                                                                                  unsigned long long *flop_counter,
                                                                                  unsigned long long *dbg)
{
    unsigned long long stamp_ = dbg ? __builtin_amdgcn_s_memtime() : 0;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int ldp = nb + 2; // leading dimensions padded by one 16-byte slot: MFMA operand reads stay conflict-free
    double *P = reinterpret_cast<double *>(smem_raw); // P[c * ldp + r]: column c (0..15) of the panel, row r (absolute)
    double *S = P + GETRF_PANEL * ldp;                // S[k * ldp + c]: row k (0..15) of the strip, column c (absolute)
    double *Rb = S + GETRF_PANEL * ldp;               // Rb[kk * 16 + c]: pivot row kk of the panel, published per step
    u32 *sLcp = reinterpret_cast<u32 *>(Rb + GETRF_PANEL * GETRF_PANEL); // column pointer of the lower half (nb + 1 entries)
    u32 *sUrp = sLcp + nb + 1;                                           // row pointer of the upper half
    const GetrfTaskD T = tasks[blockIdx.x];
    double *__restrict__ D = reinterpret_cast<double *>(T.dense);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwaves = GETRF_BLOCKED_THREADS / 64;

    // dense image: zero, then scatter both halves (unless the caller hands over a current dense mirror)
    if (!T.preloaded)
    {
        for (int i = tid; i < nb * nb / 2; i += GETRF_BLOCKED_THREADS)
            reinterpret_cast<double2 *>(D)[i] = make_double2(0.0, 0.0);
    }
    for (int i = tid; i <= nb; i += GETRF_BLOCKED_THREADS)
    {
        sLcp[i] = T.lcp[i];
        sUrp[i] = T.urp[i];
    }
    __syncthreads();
    // scatter / gather run flat over the nonzeros (coalesced, many loads in flight); the owning column (row) of a
    // position comes from a binary search in the LDS copy of the pointer array
    const u32 nnzL = sLcp[nb], nnzU = sUrp[nb];
    unsigned long long ops = 0;
    if (!T.preloaded)
    {
        for (u32 p = tid; p < nnzL; p += GETRF_BLOCKED_THREADS)
            D[(size_t)owner_of(sLcp, nb, p) * nb + T.lri[p]] = T.lval[p];
        for (u32 p = tid; p < nnzU; p += GETRF_BLOCKED_THREADS)
            D[(size_t)T.uci[p] * nb + owner_of(sUrp, nb, p)] = T.uval[p];
    }
    for (int c = tid; c < nb; c += GETRF_BLOCKED_THREADS)
    {
        // structural flop count of the sparse algorithm (what the reference counts, src/pangulu_kernel_interface.c:4-82)
        const u32 nl = sLcp[c + 1] - sLcp[c], nu = sUrp[c + 1] - sUrp[c];
        if (nu > 0)
            ops += (unsigned long long)nl * (1ull + 2ull * (nu - 1));
    }
    __syncthreads();
    GETRF_STAMP(0)

    for (int j0 = 0; j0 < nb; j0 += GETRF_PANEL)
    {
        const int jt = j0 + GETRF_PANEL; // first trailing row/column
        // ---- panel: thread t < nb - j0 owns row j0 + t of the 16 panel columns in registers ---------------------
        // Per pivot the owner of the pivot row publishes it through LDS (one barrier), every row below scales its
        // own L entry and updates its own 15 registers: no LDS traffic besides the 16-value broadcast.
        const int myrow = j0 + tid;
        const bool row_thread = tid < GETRF_BLOCKED_ROWS && myrow < nb;
        double x[GETRF_PANEL];
        if (row_thread)
        {
#pragma unroll
            for (int c = 0; c < GETRF_PANEL; c++)
                x[c] = D[(size_t)(j0 + c) * nb + myrow];
        }
        GETRF_STAMP(1)
#pragma unroll
        for (int kk = 0; kk < GETRF_PANEL; kk++)
        {
            if (tid == kk)
            {
#pragma unroll
                for (int c = 0; c < GETRF_PANEL; c++)
                    Rb[kk * GETRF_PANEL + c] = x[c];
            }
            __syncthreads();
            const int k = j0 + kk;
            if (sLcp[k] != sLcp[k + 1] && row_thread && myrow > k && x[kk] != 0.0)
            {
                const double l = x[kk] / clamp_pivot(Rb[kk * GETRF_PANEL + kk]);
                x[kk] = l;
#pragma unroll
                for (int c = 0; c < GETRF_PANEL; c++)
                    if (c > kk)
                        x[c] = x[c] - l * Rb[kk * GETRF_PANEL + c];
            }
        }
        GETRF_STAMP(2)
        if (row_thread)
        {
#pragma unroll
            for (int c = 0; c < GETRF_PANEL; c++)
            {
                D[(size_t)(j0 + c) * nb + myrow] = x[c];
                P[c * ldp + myrow] = x[c];
            }
        }
        __syncthreads();
        GETRF_STAMP(3)
        // ---- strip: thread t < nb - jt owns column jt + t of the 16 strip rows; forward substitution with the unit
        // lower 16 x 16 tile L11 read (broadcast) from the panel image ------------------------------------------------
        if (tid < nb - jt)
        {
            const int c = jt + tid;
            double s[GETRF_PANEL];
            const double2 *src = reinterpret_cast<const double2 *>(D + (size_t)c * nb + j0);
#pragma unroll
            for (int q = 0; q < GETRF_PANEL / 2; q++)
            {
                const double2 v = src[q];
                s[2 * q] = v.x;
                s[2 * q + 1] = v.y;
            }
#pragma unroll
            for (int kk = 0; kk < GETRF_PANEL; kk++)
            {
                if (s[kk] != 0.0)
                {
#pragma unroll
                    for (int rr = 0; rr < GETRF_PANEL; rr++)
                        if (rr > kk)
                            s[rr] = s[rr] - P[kk * ldp + j0 + rr] * s[kk];
                }
            }
            double2 *dst = reinterpret_cast<double2 *>(D + (size_t)c * nb + j0);
#pragma unroll
            for (int q = 0; q < GETRF_PANEL / 2; q++)
                dst[q] = make_double2(s[2 * q], s[2 * q + 1]);
#pragma unroll
            for (int kk = 0; kk < GETRF_PANEL; kk++)
                S[kk * ldp + c] = s[kk];
        }
        __syncthreads();
        GETRF_STAMP(4)
        // ---- trailing update on the matrix cores ---------------------------------------------------------------
        const int mt = (nb - jt) / 16; // trailing tiles per dimension
        const int l15 = lane & 15, l4 = lane >> 4;
        // a wavefront takes 32 x 32 macro tiles (2 x 2 MFMA tiles sharing their operands): 16 accumulator loads in
        // flight per pass instead of 4, half the LDS operand reads per flop
        const int mm = (mt + 1) / 2;
        for (int mtile = wave; mtile < mm * mm; mtile += nwaves)
        {
            const int ri = (mtile % mm) * 2, ci = (mtile / mm) * 2;
            const int r0 = jt + ri * 16, c0 = jt + ci * 16;
            const bool hr = ri + 1 < mt, hc = ci + 1 < mt; // second row / column of tiles exists
            const int r1 = hr ? r0 + 16 : r0, c1 = hc ? c0 + 16 : c0;
            // operands: A[i = l15][k = l4] = -U(k, c+i);  B[k = l4][j = l15] = L(r+j, k)
            double a0[4], a1[4], b0[4], b1[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
            {
                a0[q] = -S[(q * 4 + l4) * ldp + c0 + l15];
                a1[q] = -S[(q * 4 + l4) * ldp + c1 + l15];
                b0[q] = P[(q * 4 + l4) * ldp + r0 + l15];
                b1[q] = P[(q * 4 + l4) * ldp + r1 + l15];
            }
            const bool za0 = !__any((a0[0] != 0.0) | (a0[1] != 0.0) | (a0[2] != 0.0) | (a0[3] != 0.0));
            const bool za1 = !hc || !__any((a1[0] != 0.0) | (a1[1] != 0.0) | (a1[2] != 0.0) | (a1[3] != 0.0));
            const bool zb0 = !__any((b0[0] != 0.0) | (b0[1] != 0.0) | (b0[2] != 0.0) | (b0[3] != 0.0));
            const bool zb1 = !hr || !__any((b1[0] != 0.0) | (b1[1] != 0.0) | (b1[2] != 0.0) | (b1[3] != 0.0));
            // tile (x, y) = rows r_x, columns c_y; skipped when its L rows or U columns are all zero (uniform)
            const bool d00 = !(zb0 || za0), d10 = !(zb1 || za0), d01 = !(zb0 || za1), d11 = !(zb1 || za1);
            v4f64 t00 = {0.0, 0.0, 0.0, 0.0}, t10 = t00, t01 = t00, t11 = t00;
            // accumulator register g of lane l is D(r + l15, c + l4 + 4g)
#pragma unroll
            for (int g = 0; g < 4; g++)
            {
                if (d00)
                    t00[g] = D[(size_t)(c0 + l4 + 4 * g) * nb + r0 + l15];
                if (d10)
                    t10[g] = D[(size_t)(c0 + l4 + 4 * g) * nb + r1 + l15];
                if (d01)
                    t01[g] = D[(size_t)(c1 + l4 + 4 * g) * nb + r0 + l15];
                if (d11)
                    t11[g] = D[(size_t)(c1 + l4 + 4 * g) * nb + r1 + l15];
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
            {
                if (d00)
                    t00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b0[q], t00, 0, 0, 0);
                if (d10)
                    t10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b1[q], t10, 0, 0, 0);
                if (d01)
                    t01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b0[q], t01, 0, 0, 0);
                if (d11)
                    t11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b1[q], t11, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; g++)
            {
                if (d00)
                    D[(size_t)(c0 + l4 + 4 * g) * nb + r0 + l15] = t00[g];
                if (d10)
                    D[(size_t)(c0 + l4 + 4 * g) * nb + r1 + l15] = t10[g];
                if (d01)
                    D[(size_t)(c1 + l4 + 4 * g) * nb + r0 + l15] = t01[g];
                if (d11)
                    D[(size_t)(c1 + l4 + 4 * g) * nb + r1 + l15] = t11[g];
            }
        }
        __syncthreads();
        GETRF_STAMP(5)
    }

    if (T.defer_gather)
    {
        // the factors stay in the dense image; the diagonal tiles are saved behind the mirror (values + occupancy map)
        // because diag_tile_inverse_kernel replaces them by their inverses before the sparsify job reads the image
        double *__restrict__ saved = D + (size_t)nb * nb + MIRROR_MAP_BYTES / sizeof(double);
        for (int i = tid; i < nb * 16; i += GETRF_BLOCKED_THREADS)
        {
            const int pt = i >> 8, cc = (i >> 4) & 15, rr = i & 15;
            saved[i] = D[(size_t)(16 * pt + cc) * nb + 16 * pt + rr];
        }
    }
    else
    {
        // gather the factors back into the sparse record: four entries per thread and pass, so that the index loads, the
        // searches and the reads of D of different entries overlap (one entry at a time is a chain of three dependent L2
        // round trips per entry: 56 of the kernel's 330 us)
        constexpr int GU = 4;
        for (u32 p0 = tid; p0 < nnzL; p0 += GU * GETRF_BLOCKED_THREADS)
        {
            u32 r[GU];
            double v[GU];
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * GETRF_BLOCKED_THREADS;
                r[u] = p < nnzL ? T.lri[p] : 0u;
            }
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * GETRF_BLOCKED_THREADS;
                v[u] = p < nnzL ? D[(size_t)owner_of(sLcp, nb, p) * nb + r[u]] : 0.0;
            }
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * GETRF_BLOCKED_THREADS;
                if (p < nnzL)
                    T.lval[p] = v[u];
            }
        }
        for (u32 p0 = tid; p0 < nnzU; p0 += GU * GETRF_BLOCKED_THREADS)
        {
            u32 c[GU];
            double v[GU];
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * GETRF_BLOCKED_THREADS;
                c[u] = p < nnzU ? T.uci[p] : 0u;
            }
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * GETRF_BLOCKED_THREADS;
                v[u] = p < nnzU ? D[(size_t)c[u] * nb + owner_of(sUrp, nb, p)] : 0.0;
            }
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * GETRF_BLOCKED_THREADS;
                if (p < nnzU)
                    T.uval[p] = v[u];
            }
        }
    }
    __syncthreads();
    GETRF_STAMP(6)
    if (T.invert_tiles)
    {
        // one wavefront per diagonal tile, 16 x 17 doubles of the (now free) panel image each
        double(*Tw)[17] = reinterpret_cast<double(*)[17]>(smem_raw) + wave * 16;
        for (int p0 = 0; p0 < nb / 16; p0 += nwaves)
            invert_diag_tile(D, nb, p0 + wave, Tw, lane, p0 + wave < nb / 16, []()
                             { __syncthreads(); });
    }
    ops = wave_sum(ops);
    if (lane == 0 && ops)
        atomicAdd(flop_counter, ops);
}
// -----------------------------------------------------------------------------------------------------------------
// Blocked GETRF with look-ahead inside the block (PANGULU_HIP_GETRF_LOOKAHEAD=1): while twelve wavefronts apply panel j
// to the trailing block, the other four first update the tiles panel j+1 and its strip consist of, then eliminate panel
// j+1 and solve its strip -- the two latency-bound phases of a panel step run beside the trailing update of the previous
// one.  The four synchronise among themselves through an LDS counter.
// -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void getrf_lookahead_f64_kernel(const GetrfTaskD *__restrict__ tasks, int nb,
                                                                                  unsigned long long *flop_counter,
                                                                                  unsigned long long *dbg)
{
    unsigned long long stamp_ = dbg ? __builtin_amdgcn_s_memtime() : 0;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int ldp = nb + 2; // leading dimensions padded by one 16-byte slot: MFMA operand reads stay conflict-free
    // two images of the panel and of the strip: while the trailing update reads the current ones, the look-ahead
    // wavefronts build the next ones
    double *Pb = reinterpret_cast<double *>(smem_raw);  // Pb[(buf * 16 + c) * ldp + r]: column c of the panel, row r (absolute)
    double *Sb = Pb + 2 * GETRF_PANEL * ldp;            // Sb[(buf * 16 + k) * ldp + c]: row k of the strip, column c (absolute)
    double *Rb = Sb + 2 * GETRF_PANEL * ldp;            // Rb[kk * 16 + c]: pivot row kk of the panel, published per step
    u32 *sLcp = reinterpret_cast<u32 *>(Rb + GETRF_PANEL * GETRF_PANEL); // column pointer of the lower half (nb + 1 entries)
    u32 *sUrp = sLcp + nb + 1;                                           // row pointer of the upper half
    unsigned *la_count = sUrp + nb + 1;                                  // arrivals at the look-ahead wavefronts' own barrier
    const GetrfTaskD T = tasks[blockIdx.x];
    double *__restrict__ D = reinterpret_cast<double *>(T.dense);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwaves = 1024 / 64;

    // dense image: zero, then scatter both halves (unless the caller hands over a current dense mirror)
    if (!T.preloaded)
    {
        for (int i = tid; i < nb * nb / 2; i += 1024)
            reinterpret_cast<double2 *>(D)[i] = make_double2(0.0, 0.0);
    }
    for (int i = tid; i <= nb; i += 1024)
    {
        sLcp[i] = T.lcp[i];
        sUrp[i] = T.urp[i];
    }
    __syncthreads();
    // scatter / gather run flat over the nonzeros (coalesced, many loads in flight); the owning column (row) of a
    // position comes from a binary search in the LDS copy of the pointer array
    const u32 nnzL = sLcp[nb], nnzU = sUrp[nb];
    unsigned long long ops = 0;
    if (!T.preloaded)
    {
        for (u32 p = tid; p < nnzL; p += 1024)
            D[(size_t)owner_of(sLcp, nb, p) * nb + T.lri[p]] = T.lval[p];
        for (u32 p = tid; p < nnzU; p += 1024)
            D[(size_t)T.uci[p] * nb + owner_of(sUrp, nb, p)] = T.uval[p];
    }
    for (int c = tid; c < nb; c += 1024)
    {
        // structural flop count of the sparse algorithm (what the reference counts, src/pangulu_kernel_interface.c:4-82)
        const u32 nl = sLcp[c + 1] - sLcp[c], nu = sUrp[c + 1] - sUrp[c];
        if (nu > 0)
            ops += (unsigned long long)nl * (1ull + 2ull * (nu - 1));
    }
    __syncthreads();
    GETRF_STAMP(0)


    constexpr int LA = 4; // look-ahead wavefronts: all row threads (nb <= 256) and all strip threads live in them
    if (tid == 0)
        *la_count = 0;
    unsigned la_target = 0;
    // barrier of the LA look-ahead wavefronts only (the others are busy with the trailing update and must not be held
    // up): arrivals are counted in LDS.  LDS operations of a wavefront execute in order, so what a wavefront wrote
    // before it arrived is visible to whoever sees its arrival; vmcnt(0) orders its global stores the same way.
    auto la_barrier = [&]()
    {
        la_target += LA;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0)
            atomicAdd(la_count, 1u);
        while (*(volatile unsigned *)la_count < la_target)
            __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    const int l15 = lane & 15, l4 = lane >> 4;

    // ---- panel j0 into image `buf`: thread t < nb - j0 owns row j0 + t of the 16 panel columns in registers.  The 16 pivot
    // rows are rows of wavefront 0: it eliminates its 64 rows on its own, pivot row by pivot row, broadcasting each from its
    // lane with v_readlane (no LDS, no barrier) and publishing it for the others; after ONE barrier (`sync`) the other row
    // wavefronts run the same 16 steps on their rows from the published rows.  Same operations in the same order per
    // row as a barrier per pivot, a sixteenth of the barriers -- which matters here, where a barrier is an LDS counter.
    auto panel = [&](int j0, int buf, auto sync)
    {
        double *P = Pb + (size_t)buf * GETRF_PANEL * ldp;
        const int myrow = j0 + tid;
        const bool row_thread = tid < GETRF_BLOCKED_ROWS && myrow < nb;
        double x[GETRF_PANEL];
#pragma unroll
        for (int c = 0; c < GETRF_PANEL; c++)
            x[c] = row_thread ? D[(size_t)(j0 + c) * nb + myrow] : 0.0;
        if (wave == 0)
        {
#pragma unroll
            for (int kk = 0; kk < GETRF_PANEL; kk++)
            {
                double u[GETRF_PANEL]; // pivot row kk (wavefront-uniform)
#pragma unroll
                for (int c = 0; c < GETRF_PANEL; c++)
                    if (c >= kk)
                    {
                        union
                        {
                            double d;
                            int w[2];
                        } v;
                        v.d = x[c];
                        v.w[0] = __builtin_amdgcn_readlane(v.w[0], kk);
                        v.w[1] = __builtin_amdgcn_readlane(v.w[1], kk);
                        u[c] = v.d;
                    }
                if (lane == kk)
                {
#pragma unroll
                    for (int c = 0; c < GETRF_PANEL; c++)
                        Rb[kk * GETRF_PANEL + c] = x[c];
                }
                const int k = j0 + kk;
                if (sLcp[k] != sLcp[k + 1] && row_thread && myrow > k && x[kk] != 0.0)
                {
                    const double l = x[kk] / clamp_pivot(u[kk]);
                    x[kk] = l;
#pragma unroll
                    for (int c = 0; c < GETRF_PANEL; c++)
                        if (c > kk)
                            x[c] = x[c] - l * u[c];
                }
            }
        }
        sync();
        if (wave != 0 && row_thread)
        {
#pragma unroll
            for (int kk = 0; kk < GETRF_PANEL; kk++)
            {
                const int k = j0 + kk;
                if (sLcp[k] != sLcp[k + 1] && x[kk] != 0.0) // (myrow > k: these rows are at least 64 below the panel's first)
                {
                    const double l = x[kk] / clamp_pivot(Rb[kk * GETRF_PANEL + kk]);
                    x[kk] = l;
#pragma unroll
                    for (int c = 0; c < GETRF_PANEL; c++)
                        if (c > kk)
                            x[c] = x[c] - l * Rb[kk * GETRF_PANEL + c];
                }
            }
        }
        if (row_thread)
        {
#pragma unroll
            for (int c = 0; c < GETRF_PANEL; c++)
            {
                D[(size_t)(j0 + c) * nb + myrow] = x[c];
                P[c * ldp + myrow] = x[c];
            }
        }
        sync();
    };
    // ---- strip of panel j0 into image `buf`: thread t < nb - jt owns column jt + t of the 16 strip rows; forward
    // substitution with the unit lower 16 x 16 tile L11 read (broadcast) from the panel image
    auto strip = [&](int j0, int buf, auto sync)
    {
        const double *P = Pb + (size_t)buf * GETRF_PANEL * ldp;
        double *S = Sb + (size_t)buf * GETRF_PANEL * ldp;
        const int jt = j0 + GETRF_PANEL;
        if (tid < nb - jt)
        {
            const int c = jt + tid;
            double s[GETRF_PANEL];
            const double2 *src = reinterpret_cast<const double2 *>(D + (size_t)c * nb + j0);
#pragma unroll
            for (int q = 0; q < GETRF_PANEL / 2; q++)
            {
                const double2 v = src[q];
                s[2 * q] = v.x;
                s[2 * q + 1] = v.y;
            }
#pragma unroll
            for (int kk = 0; kk < GETRF_PANEL; kk++)
            {
                if (s[kk] != 0.0)
                {
#pragma unroll
                    for (int rr = 0; rr < GETRF_PANEL; rr++)
                        if (rr > kk)
                            s[rr] = s[rr] - P[kk * ldp + j0 + rr] * s[kk];
                }
            }
            double2 *dst = reinterpret_cast<double2 *>(D + (size_t)c * nb + j0);
#pragma unroll
            for (int q = 0; q < GETRF_PANEL / 2; q++)
                dst[q] = make_double2(s[2 * q], s[2 * q + 1]);
#pragma unroll
            for (int kk = 0; kk < GETRF_PANEL; kk++)
                S[kk * ldp + c] = s[kk];
        }
        sync();
    };
    // ---- one 16 x 16 tile (rows r0.., columns c0..) of the trailing block minus the rank-16 product of image `buf`
    // (A[i = l15][k = l4] = -U(k, c + i), B[k = l4][j = l15] = L(r + j, k); accumulator register g of lane l is
    // D(r + l15, c + l4 + 4g)): four tiles per pass, their loads together, then the MFMAs, then the stores
    auto update_tiles4 = [&](const int (&r0)[4], const int (&c0)[4], int ntile, int buf)
    {
        const double *P = Pb + (size_t)buf * GETRF_PANEL * ldp;
        const double *S = Sb + (size_t)buf * GETRF_PANEL * ldp;
        v4f64 t[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
        {
            t[u] = (v4f64){0.0, 0.0, 0.0, 0.0};
            if (u < ntile)
            {
#pragma unroll
                for (int g = 0; g < 4; g++)
                    t[u][g] = D[(size_t)(c0[u] + l4 + 4 * g) * nb + r0[u] + l15];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (u < ntile)
            {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    t[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(-S[(q * 4 + l4) * ldp + c0[u] + l15], P[(q * 4 + l4) * ldp + r0[u] + l15], t[u], 0, 0, 0);
            }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (u < ntile)
            {
#pragma unroll
                for (int g = 0; g < 4; g++)
                    D[(size_t)(c0[u] + l4 + 4 * g) * nb + r0[u] + l15] = t[u][g];
            }
    };
    auto wg_sync = []()
    { __syncthreads(); };

    // panel 0 and its strip by everyone, as in the kernel without look-ahead
    __syncthreads(); // (la_count)
    panel(0, 0, wg_sync);
    strip(0, 0, wg_sync);
    for (int j0 = 0; j0 + GETRF_PANEL < nb; j0 += GETRF_PANEL)
    {
        const int cur = (j0 / GETRF_PANEL) & 1, nxt = cur ^ 1;
        const int jt = j0 + GETRF_PANEL; // first trailing row / column
        const int mt = (nb - jt) / 16;   // trailing tiles per dimension (>= 1)
        if (wave < LA)
        {
            // (a) the tiles the next panel and the next strip consist of: tile column 0 (mt tiles) and the rest of tile
            // row 0 (mt - 1 tiles) of the trailing block, dealt over the look-ahead wavefronts four at a time
            const int npri = 2 * mt - 1;
            for (int base = wave * 4; base < npri; base += LA * 4)
            {
                int r0[4], c0[4], n = 0;
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    const int i = base + u;
                    r0[u] = c0[u] = jt;
                    if (i < npri)
                    {
                        r0[u] = i < mt ? jt + 16 * i : jt;            // column 0: rows i
                        c0[u] = i < mt ? jt : jt + 16 * (i - mt + 1); // row 0: columns 1..
                        n = u + 1;
                    }
                }
                update_tiles4(r0, c0, n, cur);
            }
            __builtin_amdgcn_s_waitcnt(0); // (stores of the tiles before the arrival: see la_barrier)
            la_barrier();
            // (b) next panel, (c) next strip, into the other images
            panel(jt, nxt, la_barrier);
            if (jt + GETRF_PANEL < nb)
                strip(jt, nxt, la_barrier);
        }
        else
        {
            // the rest of the trailing block (tile rows and columns >= 1) on the other wavefronts, 32 x 32 macro tiles
            const double *P = Pb + (size_t)cur * GETRF_PANEL * ldp;
            const double *S = Sb + (size_t)cur * GETRF_PANEL * ldp;
            const int m1 = mt - 1, mm = (m1 + 1) / 2;
            for (int mtile = wave - LA; mtile < mm * mm; mtile += nwaves - LA)
            {
                const int ri = (mtile % mm) * 2, ci = (mtile / mm) * 2;
                const int r0 = jt + 16 + ri * 16, c0 = jt + 16 + ci * 16;
                const bool hr = ri + 1 < m1, hc = ci + 1 < m1; // second row / column of tiles exists
                const int r1 = hr ? r0 + 16 : r0, c1 = hc ? c0 + 16 : c0;
                double a0[4], a1[4], b0[4], b1[4];
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    a0[q] = -S[(q * 4 + l4) * ldp + c0 + l15];
                    a1[q] = -S[(q * 4 + l4) * ldp + c1 + l15];
                    b0[q] = P[(q * 4 + l4) * ldp + r0 + l15];
                    b1[q] = P[(q * 4 + l4) * ldp + r1 + l15];
                }
                const bool za0 = !__any((a0[0] != 0.0) | (a0[1] != 0.0) | (a0[2] != 0.0) | (a0[3] != 0.0));
                const bool za1 = !hc || !__any((a1[0] != 0.0) | (a1[1] != 0.0) | (a1[2] != 0.0) | (a1[3] != 0.0));
                const bool zb0 = !__any((b0[0] != 0.0) | (b0[1] != 0.0) | (b0[2] != 0.0) | (b0[3] != 0.0));
                const bool zb1 = !hr || !__any((b1[0] != 0.0) | (b1[1] != 0.0) | (b1[2] != 0.0) | (b1[3] != 0.0));
                const bool d00 = !(zb0 || za0), d10 = !(zb1 || za0), d01 = !(zb0 || za1), d11 = !(zb1 || za1);
                v4f64 t00 = {0.0, 0.0, 0.0, 0.0}, t10 = t00, t01 = t00, t11 = t00;
#pragma unroll
                for (int g = 0; g < 4; g++)
                {
                    if (d00)
                        t00[g] = D[(size_t)(c0 + l4 + 4 * g) * nb + r0 + l15];
                    if (d10)
                        t10[g] = D[(size_t)(c0 + l4 + 4 * g) * nb + r1 + l15];
                    if (d01)
                        t01[g] = D[(size_t)(c1 + l4 + 4 * g) * nb + r0 + l15];
                    if (d11)
                        t11[g] = D[(size_t)(c1 + l4 + 4 * g) * nb + r1 + l15];
                }
#pragma unroll
                for (int q = 0; q < 4; q++)
                {
                    if (d00)
                        t00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b0[q], t00, 0, 0, 0);
                    if (d10)
                        t10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b1[q], t10, 0, 0, 0);
                    if (d01)
                        t01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b0[q], t01, 0, 0, 0);
                    if (d11)
                        t11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b1[q], t11, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; g++)
                {
                    if (d00)
                        D[(size_t)(c0 + l4 + 4 * g) * nb + r0 + l15] = t00[g];
                    if (d10)
                        D[(size_t)(c0 + l4 + 4 * g) * nb + r1 + l15] = t10[g];
                    if (d01)
                        D[(size_t)(c1 + l4 + 4 * g) * nb + r0 + l15] = t01[g];
                    if (d11)
                        D[(size_t)(c1 + l4 + 4 * g) * nb + r1 + l15] = t11[g];
                }
            }
        }
        __syncthreads(); // the trailing block is up to date, the next panel and strip images are complete
    }

    if (T.defer_gather)
    {
        // the factors stay in the dense image; the diagonal tiles are saved behind the mirror (values + occupancy map)
        // because diag_tile_inverse_kernel replaces them by their inverses before the sparsify job reads the image
        double *__restrict__ saved = D + (size_t)nb * nb + MIRROR_MAP_BYTES / sizeof(double);
        for (int i = tid; i < nb * 16; i += 1024)
        {
            const int pt = i >> 8, cc = (i >> 4) & 15, rr = i & 15;
            saved[i] = D[(size_t)(16 * pt + cc) * nb + 16 * pt + rr];
        }
    }
    else
    {
        // gather the factors back into the sparse record: four entries per thread and pass, so that the index loads, the
        // searches and the reads of D of different entries overlap (one entry at a time is a chain of three dependent L2
        // round trips per entry: 56 of the kernel's 330 us)
        constexpr int GU = 4;
        for (u32 p0 = tid; p0 < nnzL; p0 += GU * 1024)
        {
            u32 r[GU];
            double v[GU];
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * 1024;
                r[u] = p < nnzL ? T.lri[p] : 0u;
            }
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * 1024;
                v[u] = p < nnzL ? D[(size_t)owner_of(sLcp, nb, p) * nb + r[u]] : 0.0;
            }
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * 1024;
                if (p < nnzL)
                    T.lval[p] = v[u];
            }
        }
        for (u32 p0 = tid; p0 < nnzU; p0 += GU * 1024)
        {
            u32 c[GU];
            double v[GU];
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * 1024;
                c[u] = p < nnzU ? T.uci[p] : 0u;
            }
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * 1024;
                v[u] = p < nnzU ? D[(size_t)c[u] * nb + owner_of(sUrp, nb, p)] : 0.0;
            }
    #pragma unroll
            for (int u = 0; u < GU; u++)
            {
                const u32 p = p0 + u * 1024;
                if (p < nnzU)
                    T.uval[p] = v[u];
            }
        }
    }
    __syncthreads();
    GETRF_STAMP(6)
    if (T.invert_tiles)
    {
        // one wavefront per diagonal tile, 16 x 17 doubles of the (now free) panel image each
        double(*Tw)[17] = reinterpret_cast<double(*)[17]>(smem_raw) + wave * 16;
        for (int p0 = 0; p0 < nb / 16; p0 += nwaves)
            invert_diag_tile(D, nb, p0 + wave, Tw, lane, p0 + wave < nb / 16, []()
                             { __syncthreads(); });
    }
    ops = wave_sum(ops);
    if (lane == 0 && ops)
        atomicAdd(flop_counter, ops);
}
#include "pg_hip_getrf_tiled.h"

// -----------------------------------------------------------------------------------------------------------------
// GETRF of a level's diagonal blocks and the dense TSTRF/GESSM against them in ONE launch (the reference runs them as
// dependent tasks one behind the other, src/pangulu_numeric.c:655-769; near the root of the elimination tree that chain of
// two ~200 us kernels per level IS the factorisation).  Workgroups 0 .. ngetrf-1 factorise (they are dispatched first: a
// launch hands out its workgroups in index order), every other workgroup solves two (task, 64-wide slab) items, one per
// half of its eight wavefronts, and each wavefront starts panel p of its strip when the factorisation of its diagonal
// block has published panel p: the solves end one or two panel steps behind the factorisation instead of starting there.
// Hand-off: `sc1` stores + `sc1` loads + one progress word per diagonal block (pg_hip_getrf_tiled.h, gt_publish).
// MEASURED (profiles/r03z_chase.log, shell(398), one box): correct -- same factors, residual, factor check -- and NOT faster:
// a two-in-one launch takes 363 us where the lone factorisation takes 205-215 and the level's few dense solves about 100
// behind it (the factorisation pays for the hand-off: a vmcnt(0) drain and a flag per panel step, sc1 stores, the tile
// inverses inside the loop; the solves trail it by a full step and finish their longest panel after it); the factorisation
// goes from 36.9 to 37.7 / 38.9 / 41.1 ms with at most 1 / 4 / 16 factorisations per chased launch.  OFF by default
// (PANGULU_HIP_CHASE=1, PANGULU_HIP_CHASE_MAX_GETRF); kept with its parity test as the hand-off mechanism a cheaper
// producer side could reuse.
// -----------------------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(GT_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void getrf_trsm_chase_kernel(
    const GetrfTaskD *__restrict__ gtasks, unsigned ngetrf, unsigned *__restrict__ progress, unsigned long long *flop_counter,
    const TrsmDenseTaskD *__restrict__ ttasks, const u32 *__restrict__ work, unsigned nwork)
{
    // (both roles inlined: as called functions they were slower still, 457 against 363 us per launch on shell(398))
    if (blockIdx.x < ngetrf)
    {
        const GetrfTaskD T = gtasks[blockIdx.x];
        getrf_tiled_body<true>(T, NP * 16, flop_counter, nullptr, progress + blockIdx.x);
        return;
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned w = (blockIdx.x - ngetrf) * 2u + (unsigned)(wave >> 2);
    if (w >= nwork)
        return;
    const u32 item = work[w];
    const TrsmDenseTaskD T = ttasks[item >> 2];
    trsm_dense_direct_body<NP, true>(T, (int)(item & 3u), wave & 3, (int)(threadIdx.x & 63));
}

__global__ void zero_words_kernel(unsigned *p, unsigned n)
{
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x)
        p[i] = 0u;
}
#endif

#if defined(PG_COMPLEX_PANELS)
// structural flops of the GETRFs that ran in their mirrors (what the reference counts, src/pangulu_kernel_interface.c:4-82:
// per column, entries below the diagonal x (1 + 2 x entries right of the diagonal in that row)); one workgroup per block
__global__ void getrf_flop_count_kernel(const GetrfTaskD *__restrict__ tasks, int nb, unsigned long long *counter)
{
    const GetrfTaskD T = tasks[blockIdx.x];
    unsigned long long ops = 0;
    for (int c = threadIdx.x; c < nb; c += blockDim.x)
    {
        const u32 nl = T.lcp[c + 1] - ptr0(T.lcp, c), nu = T.urp[c + 1] - ptr0(T.urp, c);
        if (nu > 0)
            ops += (unsigned long long)nl * (1ull + 2ull * (nu - 1));
    }
    ops = wave_sum(ops);
    if ((threadIdx.x & 63) == 0 && ops)
        atomicAdd(counter, ops);
}
#endif

// -----------------------------------------------------------------------------------------------------------------
// solve-side kernels (device pointers x, y), semantics of ...0100000.c:435-506
// -----------------------------------------------------------------------------------------------------------------
__global__ void spmv_kernel(int nb, const u32 *cp, const u16 *ri, const val_t *val, const val_t *x, val_t *y)
{
    // y -= A x; one thread per row would need CSR: instead one wavefront per column with atomics avoided by
    // running columns sequentially inside a single workgroup (nb is small, this is a latency kernel)
    for (int c = 0; c < nb; c++)
    {
        const val_t xc = x[c];
        for (u32 p = ptr0(cp, c) + threadIdx.x; p < cp[c + 1]; p += blockDim.x)
            y[ri[p]] = v_submul(y[ri[p]], val[p], xc);
        __syncthreads();
    }
}

__global__ void vecadd_kernel(long long n, val_t *b, const val_t *x)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
    {
#ifdef PANGULU_COMPLEX
        b[i].re += x[i].re;
        b[i].im += x[i].im;
#else
        b[i] += x[i];
#endif
    }
}

__global__ void sptrsv_kernel(int nb, const u32 *ptr, const u16 *idx, const val_t *val, val_t *x, int upper)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    val_t *xs = reinterpret_cast<val_t *>(smem_raw);
    for (int i = threadIdx.x; i < nb; i += blockDim.x)
        xs[i] = x[i];
    __syncthreads();
    if (!upper)
    {
        for (int c = 0; c < nb; c++)
        {
            const val_t xc = xs[c];
            for (u32 p = ptr[c] + threadIdx.x; p < ptr[c + 1]; p += blockDim.x)
                xs[idx[p]] = v_submul(xs[idx[p]], val[p], xc);
            __syncthreads();
        }
    }
    else
    {
        // rows from the bottom; the row's dot product is reduced by one wavefront
        for (int r = nb - 1; r >= 0; r--)
        {
            const u32 b = ptr[r], e = ptr[r + 1];
            if (b == e)
                continue;
            if (threadIdx.x < 64)
            {
#ifdef PANGULU_COMPLEX
                val_t part = v_make(0);
                for (u32 p = b + 1 + threadIdx.x; p < e; p += 64)
                {
                    val_t m = v_mul(val[p], xs[idx[p]]);
                    part.re += m.re;
                    part.im += m.im;
                }
                for (int off = 32; off > 0; off >>= 1)
                {
                    part.re += __shfl_down(part.re, off, 64);
                    part.im += __shfl_down(part.im, off, 64);
                }
#else
                val_t part = 0;
                for (u32 p = b + 1 + threadIdx.x; p < e; p += 64)
                    part += val[p] * xs[idx[p]];
                for (int off = 32; off > 0; off >>= 1)
                    part += __shfl_down(part, off, 64);
#endif
                if (threadIdx.x == 0)
                {
                    val_t d = val[b];
                    real_t dr = v_realpart(d);
                    val_t num = v_sub(xs[r], part);
                    xs[r] = ((dr < 0 ? -dr : dr) > (real_t)PANGULU_SPTRSV_TOL) ? v_div(num, d) : v_div(num, v_make((real_t)PANGULU_SPTRSV_TOL));
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < nb; i += blockDim.x)
        x[i] = xs[i];
}

// =================================================================================================================
// host side of the back-end
// =================================================================================================================
namespace
{

// -----------------------------------------------------------------------------------------------------------------
// Level-scheduled block triangular solve for pangulu_gstrs on a single rank (pangulu_platform_0201001_block_trsv).
// The reference sweeps block row by block row with one spmv / sptrsv platform call per block on the CPU platform
// (src/pangulu_sptrsv.c:24-191); here block rows whose inputs are final form a LEVEL of the block dependency graph and one
// level is two launches: one workgroup per off-diagonal block subtracts  A(row, j) x_j  from the row's segment (floating-
// point atomics), then one wavefront per block row solves with the row's diagonal half in LDS and writes the finished
// segment.  Same per-block arithmetic as ...0100000.c:435-506 (spmv, unit-lower column sweep, upper row sweep
// with the PANGULU_SPTRSV_TOL clamp); sums across blocks arrive in a different order.
// -----------------------------------------------------------------------------------------------------------------
struct SolveBlkD
{
    const u32 *cp; // CSC
    const u16 *ri;
    const val_t *val;
    u32 bcol;
    u32 brow; // destination segment
};
struct SolveRowD
{
    u32 brow, nblk;
    unsigned long long first; // into the SolveBlkD array
    const u32 *dptr;          // diagonal half: lower = strictly-lower CSC column pointer, upper = CSR row pointer (diagonal first)
    const u16 *didx;
    const val_t *dval;
};

// x_row -= A(row, j) x_j for every off-diagonal block of the level: one workgroup per block (rows near the root of the
// tree have hundreds of blocks: a workgroup per row would walk them one after the other), floating-point atomics on
// the destination segment
__global__ __launch_bounds__(256) void block_trsv_gather_kernel(const SolveBlkD *__restrict__ blks, int nb, val_t *__restrict__ x)
{
    const SolveBlkD B = blks[blockIdx.x];
    const val_t *xj = x + (size_t)B.bcol * nb;
    val_t *xr = x + (size_t)B.brow * nb;
    const int sub = threadIdx.x >> 4, l16 = threadIdx.x & 15, nsub = blockDim.x >> 4;
    for (int c = sub; c < nb; c += nsub)
    {
        const u32 p0 = ptr0(B.cp, c), p1 = B.cp[c + 1];
        if (p0 == p1)
            continue;
        const val_t xc = xj[c];
        for (u32 p = p0 + l16; p < p1; p += 16)
        {
            const val_t m = v_mul(B.val[p], xc);
#ifdef PANGULU_COMPLEX
            v_atomic_add(&xr[B.ri[p]], val_t{-m.re, -m.im});
#else
            v_atomic_add(&xr[B.ri[p]], -m);
#endif
        }
    }
}

// y_dst += A x_src for a list of blocks (factor check: t = U 1, then y = L t): one workgroup per block, 16 lanes per
// column (CSC record) or row (CSR record: upper diagonal half), floating-point atomics on y
struct SpmvBlkD
{
    const u32 *ptr;
    const u16 *idx;
    const val_t *val;
    u32 src, dst;
    u32 csr, pad_;
};
__global__ __launch_bounds__(256) void block_spmv_add_kernel(const SpmvBlkD *__restrict__ blks, int nb, const val_t *__restrict__ x, val_t *__restrict__ y)
{
    const SpmvBlkD B = blks[blockIdx.x];
    const val_t *xs = x + (size_t)B.src * nb;
    val_t *yd = y + (size_t)B.dst * nb;
    const int sub = threadIdx.x >> 4, l16 = threadIdx.x & 15, nsub = blockDim.x >> 4;
    for (int c = sub; c < nb; c += nsub)
    {
        const u32 p0 = ptr0(B.ptr, c), p1 = B.ptr[c + 1];
        if (p0 == p1)
            continue;
        if (!B.csr)
        {
            const val_t xc = xs[c];
            for (u32 p = p0 + l16; p < p1; p += 16)
                v_atomic_add(&yd[B.idx[p]], v_mul(B.val[p], xc));
        }
        else
        {
            val_t part = v_make(0);
            for (u32 p = p0 + l16; p < p1; p += 16)
            {
                const val_t m = v_mul(B.val[p], xs[B.idx[p]]);
#ifdef PANGULU_COMPLEX
                part.re += m.re;
                part.im += m.im;
#else
                part += m;
#endif
            }
            v_atomic_add(&yd[c], part); // (16 partial sums per row)
        }
    }
}

// the diagonal halves of the level's block rows: one wavefront per row, the segment in LDS
template <bool UPPER>
__global__ __launch_bounds__(64) void block_trsv_level_kernel(const SolveRowD *__restrict__ rows, int nb, val_t *__restrict__ x)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    val_t *seg = reinterpret_cast<val_t *>(smem_raw);
    const SolveRowD R = rows[blockIdx.x];
    val_t *xr = x + (size_t)R.brow * nb;
    for (int i = threadIdx.x; i < nb; i += blockDim.x)
        seg[i] = xr[i];
    __syncthreads();
    // the diagonal half, by one wavefront (LDS operations of a wavefront complete in order: no barriers in the sweep)
    if (threadIdx.x < 64)
    {
        const int lane = threadIdx.x;
        if (!UPPER)
        {
            for (int c = 0; c < nb; c++)
            {
                const u32 p0 = ptr0(R.dptr, c), p1 = R.dptr[c + 1];
                if (p0 == p1)
                    continue;
                const val_t xc = seg[c];
                for (u32 p = p0 + lane; p < p1; p += 64)
                    seg[R.didx[p]] = v_submul(seg[R.didx[p]], R.dval[p], xc);
                wave_lds_fence();
            }
        }
        else
        {
            for (int r = nb - 1; r >= 0; r--)
            {
                const u32 b = R.dptr[r], e = R.dptr[r + 1];
                if (b == e)
                    continue;
#ifdef PANGULU_COMPLEX
                val_t part = v_make(0);
                for (u32 p = b + 1 + lane; p < e; p += 64)
                {
                    const val_t m = v_mul(R.dval[p], seg[R.didx[p]]);
                    part.re += m.re;
                    part.im += m.im;
                }
                for (int off = 32; off > 0; off >>= 1)
                {
                    part.re += __shfl_down(part.re, off, 64);
                    part.im += __shfl_down(part.im, off, 64);
                }
#else
                val_t part = 0;
                for (u32 p = b + 1 + lane; p < e; p += 64)
                    part += R.dval[p] * seg[R.didx[p]];
                for (int off = 32; off > 0; off >>= 1)
                    part += __shfl_down(part, off, 64);
#endif
                if (lane == 0)
                {
                    val_t d = R.dval[b];
                    const real_t dr = v_realpart(d);
                    if (!((dr < 0 ? -dr : dr) > (real_t)PANGULU_SPTRSV_TOL))
                        d = v_make((real_t)PANGULU_SPTRSV_TOL);
                    seg[r] = v_div(v_sub(seg[r], part), d);
                }
                wave_lds_fence();
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += blockDim.x)
        xr[i] = seg[i];
}

// ---- round 4: the same two launches per level, rebuilt around where their time went (fem27(112): 540 launches, 394 ms) --------
// The gather kernel walked a block column by column, sixteen lanes a column, every entry a floating-point atomic on the row's
// 256 words in HBM (hundreds of blocks of a row near the root contend for them): 98 % of its wave cycles waiting.  The level kernel
// swept a diagonal half column by column straight from HBM: nb dependent round trips.
//  * gather: the block's entries flat over the workgroup (coalesced loads, the column of an entry by bisection in an LDS copy of the
//    column pointers), products accumulated in LDS (ds_add_f64), ONE global atomic per touched row of the segment at the end;
//  * level: the diagonal half streams through LDS in chunks of `ch` columns (rows for the upper sweep), double-buffered: three
//    wavefronts fetch chunk k + 1 while the first one sweeps chunk k out of LDS -- a dependent step costs LDS round trips, not HBM ones.
__global__ __launch_bounds__(256) void block_trsv_gather_flat_kernel(const SolveBlkD *__restrict__ blks, int nb, val_t *__restrict__ x)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    val_t *acc = reinterpret_cast<val_t *>(smem_raw);
    val_t *xs = acc + nb;
    u32 *cp = reinterpret_cast<u32 *>(xs + nb);
    const SolveBlkD B = blks[blockIdx.x];
    const val_t *xj = x + (size_t)B.bcol * nb;
    val_t *xr = x + (size_t)B.brow * nb;
    const int tid = threadIdx.x;
    for (int i = tid; i < nb; i += 256)
    {
        acc[i] = v_make(0);
        xs[i] = xj[i];
    }
    for (int i = tid; i <= nb; i += 256)
        cp[i] = i == 0 ? 0u : B.cp[i];
    __syncthreads();
    const u32 nnz = cp[nb];
    for (u32 p = (u32)tid; p < nnz; p += 256)
    {
        // column of entry p: the last c with cp[c] <= p
        int lo = 0, hi = nb;
        while (hi - lo > 1)
        {
            const int mid = (lo + hi) >> 1;
            if (cp[mid] <= p)
                lo = mid;
            else
                hi = mid;
        }
        lds_atomic_sub(&acc[B.ri[p]], v_mul(B.val[p], xs[lo]));
    }
    __syncthreads();
    for (int i = tid; i < nb; i += 256)
        v_atomic_add(&xr[i], acc[i]);
}

template <bool UPPER>
__global__ __launch_bounds__(256) void block_trsv_level_chunked_kernel(const SolveRowD *__restrict__ rows, int nb, val_t *__restrict__ x, int ch)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const size_t cap = (size_t)ch * (size_t)nb; // entries a chunk can hold
    val_t *seg = reinterpret_cast<val_t *>(smem_raw);
    val_t *bv0 = seg + nb, *bv1 = bv0 + cap;
    u32 *ptr = reinterpret_cast<u32 *>(bv1 + cap);
    u16 *bi0 = reinterpret_cast<u16 *>(ptr + nb + 2), *bi1 = bi0 + cap;
    const SolveRowD R = rows[blockIdx.x];
    val_t *xr = x + (size_t)R.brow * nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < nb; i += 256)
        seg[i] = xr[i];
    for (int i = tid; i <= nb; i += 256)
        ptr[i] = (!UPPER && i == 0) ? 0u : R.dptr[i];
    __syncthreads();
    const int nchunk = (nb + ch - 1) / ch;
    // chunk k: columns [k ch, (k + 1) ch) of the lower half in ascending order; rows [nb - (k + 1) ch, nb - k ch) of the upper half, descending
    auto lo_of = [&](int k) -> int { return UPPER ? max(0, nb - (k + 1) * ch) : k * ch; };
    auto hi_of = [&](int k) -> int { return UPPER ? nb - k * ch : min(nb, (k + 1) * ch); };
    auto fetch = [&](int k, int first, int nthr)
    {
        val_t *bv = (k & 1) ? bv1 : bv0;
        u16 *bi = (k & 1) ? bi1 : bi0;
        const u32 p0 = ptr[lo_of(k)], p1 = ptr[hi_of(k)];
        for (u32 p = p0 + (u32)first; p < p1; p += (u32)nthr)
        {
            bv[p - p0] = R.dval[p];
            bi[p - p0] = R.didx[p];
        }
    };
    fetch(0, tid, 256);
    __syncthreads();
    for (int k = 0; k < nchunk; k++)
    {
        if (wave != 0)
        {
            if (k + 1 < nchunk)
                fetch(k + 1, tid - 64, 192);
        }
        else
        {
            const val_t *bv = (k & 1) ? bv1 : bv0;
            const u16 *bi = (k & 1) ? bi1 : bi0;
            const int c0 = lo_of(k), c1 = hi_of(k);
            const u32 base = ptr[c0];
            if (!UPPER)
            {
                for (int c = c0; c < c1; c++)
                {
                    const u32 p0 = ptr[c] - base, p1 = ptr[c + 1] - base;
                    if (p0 == p1)
                        continue;
                    const val_t xc = seg[c];
                    for (u32 p = p0 + lane; p < p1; p += 64)
                        seg[bi[p]] = v_submul(seg[bi[p]], bv[p], xc);
                    wave_lds_fence();
                }
            }
            else
            {
                for (int r = c1 - 1; r >= c0; r--)
                {
                    const u32 b = ptr[r] - base, e = ptr[r + 1] - base;
                    if (b == e)
                        continue;
#ifdef PANGULU_COMPLEX
                    val_t part = v_make(0);
                    for (u32 p = b + 1 + lane; p < e; p += 64)
                    {
                        const val_t m = v_mul(bv[p], seg[bi[p]]);
                        part.re += m.re;
                        part.im += m.im;
                    }
                    for (int off = 32; off > 0; off >>= 1)
                    {
                        part.re += __shfl_down(part.re, off, 64);
                        part.im += __shfl_down(part.im, off, 64);
                    }
#else
                    val_t part = 0;
                    for (u32 p = b + 1 + lane; p < e; p += 64)
                        part += bv[p] * seg[bi[p]];
                    for (int off = 32; off > 0; off >>= 1)
                        part += __shfl_down(part, off, 64);
#endif
                    if (lane == 0)
                    {
                        val_t d = bv[b];
                        const real_t dr = v_realpart(d);
                        if (!((dr < 0 ? -dr : dr) > (real_t)PANGULU_SPTRSV_TOL))
                            d = v_make((real_t)PANGULU_SPTRSV_TOL);
                        seg[r] = v_div(v_sub(seg[r], part), d);
                    }
                    wave_lds_fence();
                }
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < nb; i += 256)
        xr[i] = seg[i];
}

struct DiagAux // column view of a diagonal block's upper (CSR) half, built on first use
{
    u32 *d_cp = nullptr;
    u16 *d_ri = nullptr;
    u32 *d_vi = nullptr;
    u32 nnz = 0;
    u32 brow = 0;
};

struct Ring // descriptor staging in pinned host memory that the kernels read in place, reused segment by segment
{
    static const int NSEG = 32;
    std::vector<int> pending; // segments handed to kernels since the last event record
    size_t seg_bytes = 0;
    char *h = nullptr, *d = nullptr;
    hipEvent_t ev[NSEG];
    bool used[NSEG];
    int cur = 0;
};

struct EventPair
{
    hipEvent_t a, b;
    int cls;
    unsigned long long tag[3]; // per-launch log (PANGULU_HIP_LAUNCH_LOG): workgroups, tasks, live 128 x 128 x 16 slab steps
};

struct Backend
{
    bool ready = false;
    int device = 0;
    hipStream_t stream = nullptr;
    bool bulk_streams_masked = false; // stream / stream2 leave PANGULU_HIP_RESERVED_CUS CUs to the GETRF stream
    hipStream_t stream2 = nullptr; // side stream: the MFMA update kernel runs beside the LDS update kernel
    hipStream_t stream3 = nullptr; // second side stream: GETRFs of a batch run beside its TSTRF/GESSM solves
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fork3 = nullptr, ev_join3 = nullptr;
    bool getrf_join_pending = false;
    // Records stream: the sparse record stays the authoritative form of every finished block, but the dense kernels
    // of the following steps read mirrors and LU images only.  The sparsify jobs behind the dense solves and behind
    // the blocked GETRF run here, beside whatever comes next; everything that reads or rewrites sparse records (the
    // LDS update kernel, sparse solves, densify, copies to the host, markers, synchronize) joins it first.
    hipStream_t stream_rec = nullptr;
    hipEvent_t ev_rec_fork = nullptr, ev_rec = nullptr;
    std::atomic<bool> rec_dirty{false};
    // Background stream (round 3): in a call that carries diagonal factorisations AND updates (the scheduler's look-ahead:
    // the GETRFs of the next level(s) together with every update queued anywhere), the updates are the trailing-matrix
    // work of the previous level and nothing on the critical path -- next panel's updates, GETRF, panel solves -- depends
    // on them.  They go to this stream and the main stream does NOT join at the end of the call: the solves of the next
    // panel (the following call) run beside them instead of behind them (fem27(112): the dense solves ran ALONE on the
    // device for 65 of 974 ms).  The destinations of the launches in flight are remembered; the first later call that
    // touches one of them -- as destination or operand -- makes the main stream wait first (join_background).
    // MEASURED (fem27(112), one box each): the overlap is there -- dense solves exclusive 65 -> 17 ms, GETRF 13 -> 4, two or more
    // classes at once 81 -> 245 ms.  With the scheduler in the loop the factorisation did not get faster (954.7 against 940.7 ms:
    // the extra call per level cost the host-bound run more than the overlap returned); replayed from the static schedule it
    // does: 873.1 against 887.6 ms, shell(398) 39.05 against 39.72.  On by default since then.
    hipStream_t stream_bg = nullptr;
    hipEvent_t ev_bg_fork = nullptr, ev_bg_done = nullptr;
    bool bg_active = false;
    std::unordered_set<const void *> bg_tiles;
    long long opt_background_updates = 1; // PANGULU_HIP_BACKGROUND_UPDATES=0 / option 14
    // dense-front kernel (pg_hip_front.h) for the (destination, tile) pairs all of whose queued updates have every 16 x 16
    // piece live: LDS stages of its operand pipeline (2, 3 or 4; 0 = off, everything through the general kernel)
    long long opt_front_stages = 2; // PANGULU_HIP_FRONT_STAGES / option 15: 1 = inside the general launch (no step list), 2..4 = own kernel
    long long opt_front_min_wgs = 8192; // PANGULU_HIP_FRONT_MIN_WGS: ... from this many qualifying workgroups in a launch on (sweep: fem27(112) 842.8 / 845.0 / 849.0 ms at 8192 / 2048 / never)
    long long opt_front_unit = 1;   // PANGULU_HIP_FRONT_UNIT: consecutive destinations of the front launch that share an XCD
    // general MFMA update kernel: 0 = round 2's (register staging, contiguous sub-tiles; pg_hip_dense.h), 1 / 3 / 4 = the
    // LDS-DMA pipeline with 2 / 3 / 4 stages and strided piece ownership (ssssm_tiles_f64_kernel, pg_hip_front.h), 2 = its
    // two-stage form with the per-step fixed cost taken out of the chain (ssssm_tilesv_f64_kernel)
    long long opt_tiles_stages = 2; // PANGULU_HIP_TILES_STAGES / option 16
    long long opt_tiles_unit = 1;   // PANGULU_HIP_TILES_UNIT: consecutive destinations of the general launch that share an XCD
    unsigned long long front_workgroups = 0, general_workgroups = 0;
    long long opt_records_stream = 1; // PANGULU_HIP_RECORDS_STREAM=0: sparsify on the main stream as before
    int nb_cfg = 0;
    // Bumped whenever a process-global resource that recorded launches point into is freed or re-assigned (the GETRF scratch,
    // the mirror pool, the chase's progress words): a recorded schedule is only replayed under the generation it ended in.
    unsigned long long generation = 0;
    // options
    long long opt_host_mirror = 1;
    long long opt_dense_permille = 2; // (10 until the end of round 2, 5 until round 3's sweep on replayed runs: fem27(112) 887.8 / 892.1 / 906.8 ms at 2 / 5 / 10, shell(398) 37.9 / 38.5 / 39.2)
    long long opt_profile = 0;
    long long opt_assume_independent = 0;
    long long opt_getrf_strict = 0;
    long long opt_count_flops = 1;
    long long opt_group_chunk = 8;
    long long opt_small_launch_tasks = 2048;
    long long opt_trsm_dense_permille = 5; // (round 3 sweep: shell(398) 38.2 / 38.5 / 39.4 ms at 5 / 10 / 30, fem27(112) indifferent)
    long long opt_two_streams = 1;
    double mfma_flops_executed = 0;
    // resources
    Ring ring;
    unsigned *d_progress = nullptr;        // progress words of the GETRF -> dense-solve chase (one per held factorisation task)
    size_t progress_next = 0;
    unsigned long long chase_launches = 0, chase_solves = 0;
    unsigned long long zgetrf_tasks = 0; // complex types: diagonal blocks factorised in their mirrors
    unsigned long long *d_flops = nullptr; // [6]
    val_t *getrf_scratch = nullptr;
    int getrf_scratch_slots = 0;
    std::unordered_map<const void *, DiagAux> diag_aux;
    // stats
    pangulu_hip_stats_t stats;
    std::vector<EventPair> pending_events;
    std::vector<hipEvent_t> event_pool;
    std::mutex mutex;
};

Backend B;

// ---------------------------------------------------------------------------------------------------------------
// Static schedule (round 3).  For one rank the sequence of launches of a factorisation -- kernels, grids, descriptor
// contents, stream forks and joins -- is a pure function of the block pattern and the options: nothing in it depends
// on values or on timing (one launcher thread issues everything in the scheduler's order).  The first pangulu_gstrf on a
// handle therefore RECORDS every launch and stream operation it issues (a closure each; the descriptor segments they
// read are kept instead of recycled), and every later pangulu_gstrf on that handle with the same options REPLAYS the
// list: no scheduler, no descriptor building, no host work per task -- about three thousand closures for the
// Serena-class matrix instead of 2.8 million tasks.  pangulu_platform_0201001_schedule() is the control call.
// ---------------------------------------------------------------------------------------------------------------
struct Recorder
{
    int mode = 0; // 1: recording while executing; 2: recording only (dry run of the scheduler at pangulu_init: nothing is launched)
    bool valid = false;
    const void *owner = nullptr;
    unsigned long long signature = 0;
    std::vector<std::function<void()>> ops;
    // Descriptor segments of the recorded launches.  While recording, the kernels read them in place from pinned host memory
    // (h, device-visible at d) like every other run; the REPLAYS read a copy in HBM (twin), made once when the recording ends:
    // a workgroup's first two dependent reads -- its work item, its task descriptors -- then cost an L2/HBM round trip instead
    // of two trips to host memory.  The closures are built with the twin addresses (rec_xl), the pinned originals are freed.
    struct Seg
    {
        char *h, *d, *twin;
        size_t cap;
    };
    std::vector<Seg> segs;
    size_t descriptor_bytes = 0;
    // what else the closures depend on: the block order and the generation of the back-end's shared resources when the list
    // was complete (B.generation)
    int nb = 0;
    unsigned long long generation = 0;
    // host-side counters of ONE factorisation (launches, tasks, algorithmic bytes, workgroup counts): taken as the difference
    // over the recording, added by every replay; a dry run (mode 2) launched nothing and leaves the live counters as they were
    pangulu_hip_stats_t stats_before, stats_delta;
    unsigned long long wgs_before[4] = {0, 0, 0, 0}, wgs_delta[4] = {0, 0, 0, 0}; // front, general, chase launches, chase solves
};
Recorder REC;

// GETRF -> dense-solve chase (recorded schedules only).  A launch of the tiled GETRF on the main stream is HELD until the next
// platform call: if that call is the level's dense TSTRF/GESSM against exactly these diagonal blocks, both go out as ONE launch
// (getrf_trsm_chase_kernel: the solves of panel p start when the factorisation has published panel p); anything else launches
// the held factorisation first, as it was.  `hold` keeps the preparatory launches of launch_trsm (densify of the panel blocks:
// independent of the factorisation) from doing that.
#define PROGRESS_WORDS 8192
struct PendingGetrf
{
    bool active = false, hold = false;
    int nb = 0;
    size_t take = 0;
    const void *d_tasks = nullptr;       // GetrfTaskD * (device view)
    unsigned *d_progress = nullptr;      // one word per task
    std::vector<const double *> images;  // LU images the held factorisation will leave, in task order
    std::function<void()> plain;         // the launch as it would have been
    std::function<void()> post;          // what follows the launch (record-stream fork, deferred sparsify jobs, statistics)
};
PendingGetrf PEND;
inline void flush_pending_getrf()
{
    if (!PEND.active || PEND.hold)
        return;
    PEND.active = false;
    PEND.plain();
    PEND.post();
    PEND.plain = nullptr;
    PEND.post = nullptr;
}

// PEND is back-end state like everything else: entry points that do not hold B.mutex anyway take it for the flush
inline void flush_pending_getrf_locked();

// a kernel argument as the replay will pass it: pointers into a recorded descriptor segment move to the segment's HBM twin
template <class T>
inline T rec_xl(T v)
{
    if constexpr (std::is_pointer<T>::value)
    {
        const char *p = reinterpret_cast<const char *>(v);
        for (const Recorder::Seg &sg : REC.segs)
            if (p >= sg.d && p < sg.d + sg.cap)
                return reinterpret_cast<T>(const_cast<char *>(sg.twin + (p - sg.d)));
    }
    return v;
}

template <class K, class... A>
inline void pg_launch(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, A... args)
{
    flush_pending_getrf();
    if (REC.mode != 0)
    {
        auto targs = std::make_tuple(rec_xl(args)...);
        REC.ops.emplace_back([=]()
                             { std::apply([&](auto... a)
                                          { hipLaunchKernelGGL(kernel, grid, block, (unsigned)shmem, st, a...); },
                                          targs); });
        if (REC.mode == 2)
            return;
    }
    hipLaunchKernelGGL(kernel, grid, block, (unsigned)shmem, st, args...);
}
#define PG_LAUNCH(kernel_, grid_, block_, shmem_, stream_, ...) pg_launch(kernel_, grid_, block_, shmem_, stream_, __VA_ARGS__)

inline void pg_event_record(hipEvent_t e, hipStream_t s)
{
    flush_pending_getrf();
    if (REC.mode != 0)
        REC.ops.emplace_back([e, s]() { HIP_CHECK(hipEventRecord(e, s)); });
    if (REC.mode != 2)
        HIP_CHECK(hipEventRecord(e, s));
}
inline void pg_stream_wait(hipStream_t s, hipEvent_t e)
{
    flush_pending_getrf();
    if (REC.mode != 0)
        REC.ops.emplace_back([e, s]() { HIP_CHECK(hipStreamWaitEvent(s, e, 0)); });
    if (REC.mode != 2)
        HIP_CHECK(hipStreamWaitEvent(s, e, 0));
}

inline void flush_pending_getrf_locked()
{
    if (!PEND.active) // (only ever set under the mutex by the thread that launches; a stale read here just skips a no-op)
        return;
    std::lock_guard<std::mutex> g(B.mutex);
    flush_pending_getrf();
}

void ensure_ready()
{
    if (B.ready)
        return;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
    {
        fprintf(stderr, "[PanguLU-AMD ERROR] no HIP device available (%s); the GPU_HIP platform has no CPU fallback\n",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        exit(EXIT_FAILURE);
    }
    HIP_CHECK(hipSetDevice(B.device));
    // Optional (PANGULU_HIP_RESERVED_CUS=n, default 0 = off): the bulk streams (updates, solves, mirror maintenance) leave n
    // CUs alone -- mask bit i is CU i / 8 of XCD i % 8 (tools/experiments/cu_mask_probe.hip) -- and the GETRF stream
    // (stream3) sees all of them.  A GETRF workgroup needs 139 KB of LDS, i.e. a CU to itself, and an update launch that is
    // still handing out workgroups never leaves one empty: the factorisations of the upper tree levels took 240-470 us
    // beside such a launch against 205 us alone (tools/launch_size_histogram.py).  Measured with n = 8: GETRF time 15.3 ->
    // 13.4 ms (bench matrix) and 58 -> 25 ms (fem27(80)), but the update kernel lost more than the 3 % of CUs it gave up
    // (fem27(80): 126 -> 137 ms) and the factorisation did not get faster (47.3 vs 45.4-47.7 ms; 187 vs 175 ms): off.
    {
        long reserved = 0;
        if (const char *e = getenv("PANGULU_HIP_RESERVED_CUS"))
            reserved = atol(e);
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, B.device));
        const int ncu = prop.multiProcessorCount;
        if (reserved > 0 && reserved < ncu / 2 && ncu % 32 == 0)
        {
            std::vector<uint32_t> mask((size_t)ncu / 32, 0xFFFFFFFFu);
            for (int i = ncu - (int)reserved; i < ncu; i++)
                mask[(size_t)i / 32] &= ~(1u << (i % 32));
            HIP_CHECK(hipExtStreamCreateWithCUMask(&B.stream, (uint32_t)mask.size(), mask.data()));
            HIP_CHECK(hipExtStreamCreateWithCUMask(&B.stream2, (uint32_t)mask.size(), mask.data()));
            B.bulk_streams_masked = true;
        }
        else
        {
            HIP_CHECK(hipStreamCreateWithFlags(&B.stream, hipStreamNonBlocking));
            HIP_CHECK(hipStreamCreateWithFlags(&B.stream2, hipStreamNonBlocking));
        }
    }
    HIP_CHECK(hipEventCreateWithFlags(&B.ev_fork, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&B.ev_join, hipEventDisableTiming));
    HIP_CHECK(hipStreamCreateWithFlags(&B.stream3, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&B.ev_fork3, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&B.ev_join3, hipEventDisableTiming));
    HIP_CHECK(hipStreamCreateWithFlags(&B.stream_rec, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&B.ev_rec_fork, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&B.ev_rec, hipEventDisableTiming));
    if (const char *e = getenv("PANGULU_HIP_RECORDS_STREAM"))
        B.opt_records_stream = atol(e);
    HIP_CHECK(hipStreamCreateWithFlags(&B.stream_bg, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&B.ev_bg_fork, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&B.ev_bg_done, hipEventDisableTiming));
    if (const char *e = getenv("PANGULU_HIP_BACKGROUND_UPDATES"))
        B.opt_background_updates = atol(e);
    if (const char *e = getenv("PANGULU_HIP_FRONT_STAGES"))
        B.opt_front_stages = atol(e);
    if (const char *e = getenv("PANGULU_HIP_FRONT_UNIT"))
        B.opt_front_unit = atol(e);
    if (const char *e = getenv("PANGULU_HIP_FRONT_MIN_WGS"))
        B.opt_front_min_wgs = atol(e);
    if (const char *e = getenv("PANGULU_HIP_TILES_STAGES"))
        B.opt_tiles_stages = atol(e);
    if (const char *e = getenv("PANGULU_HIP_GROUP_CHUNK"))
        B.opt_group_chunk = atol(e);
    if (const char *e = getenv("PANGULU_HIP_TILES_UNIT"))
        B.opt_tiles_unit = atol(e);
    if (const char *e = getenv("PANGULU_HIP_DENSE_PERMILLE"))
        B.opt_dense_permille = atol(e);
    if (const char *e = getenv("PANGULU_HIP_TRSM_DENSE_PERMILLE"))
        B.opt_trsm_dense_permille = atol(e);
    if (const char *e = getenv("PANGULU_HIP_SMALL_LAUNCH_TASKS"))
        B.opt_small_launch_tasks = atol(e);
    // Descriptors are written once by the host and read once per workgroup: the kernels read them straight from
    // pinned host memory (non-coherent, so the device L2 may cache them) instead of waiting for a staging copy per
    // launch (rocprofv3 showed ~1900 blit dispatches, ~50 ms, per factorisation of the bench matrix).
    B.ring.seg_bytes = (size_t)8 << 20;
    HIP_CHECK(hipHostMalloc((void **)&B.ring.h, B.ring.seg_bytes * Ring::NSEG, hipHostMallocNonCoherent | hipHostMallocMapped));
    HIP_CHECK(hipHostGetDevicePointer((void **)&B.ring.d, B.ring.h, 0));
    for (int i = 0; i < Ring::NSEG; i++)
    {
        HIP_CHECK(hipEventCreateWithFlags(&B.ring.ev[i], hipEventDisableTiming));
        B.ring.used[i] = false;
    }
    HIP_CHECK(hipMalloc((void **)&B.d_flops, sizeof(unsigned long long) * 16)); // [0..7] flop counters, [8..15] debug stamps
    HIP_CHECK(hipMemset(B.d_flops, 0, sizeof(unsigned long long) * 16));
    memset(&B.stats, 0, sizeof(B.stats));
    B.ready = true;
}

// a staging segment: host pointer to fill, device pointer the kernels will read after commit()
struct Segment
{
    char *h, *d;
    size_t cap, used;
    int index;
    template <typename T>
    T *alloc(size_t count, T **dev)
    {
        size_t off = (used + 15) & ~(size_t)15;
        if (off + sizeof(T) * count > cap)
            return nullptr;
        used = off + sizeof(T) * count;
        *dev = reinterpret_cast<T *>(d + off);
        return reinterpret_cast<T *>(h + off);
    }
};

// host-side cost of preparing launches (PANGULU_HIP_HOST_TIMING=1 prints it with every get_stats(reset))
double g_host_seconds[6] = {0, 0, 0, 0, 0, 0}; // 0 ssssm, 1 trsm, 2 getrf, 3 mirror jobs, 4 waiting for a staging segment, 5 whole calls
struct HostTimer
{
    int k;
    std::chrono::steady_clock::time_point t0;
    explicit HostTimer(int k_) : k(k_), t0(std::chrono::steady_clock::now()) {}
    ~HostTimer() { g_host_seconds[k] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// Record, behind everything launched so far, that the committed segments may be reused.  Must be called AFTER the
// kernels reading those segments have been launched (an event recorded earlier would let the host overwrite a
// segment a queued kernel has yet to read).
void release_pending_segments(hipStream_t on = nullptr)
{
    Ring &r = B.ring;
    for (int i : r.pending)
    {
        HIP_CHECK(hipEventRecord(r.ev[i], on ? on : B.stream));
        r.used[i] = true;
    }
    r.pending.clear();
}

// sparse records are about to be read or rewritten on stream s: wait for the sparsify jobs of the records stream
void join_records(hipStream_t s)
{
    if (!B.rec_dirty.load(std::memory_order_acquire))
        return;
    pg_stream_wait(s, B.ev_rec);
    if (s == B.stream)
        B.rec_dirty.store(false, std::memory_order_release);
}

// stream s is about to touch blocks that update launches on the background stream may still be writing (or: everything
// queued so far has to be complete behind s)
void join_background(hipStream_t s)
{
    if (!B.bg_active)
        return;
    pg_stream_wait(s, B.ev_bg_done);
    if (s == B.stream)
    {
        B.bg_active = false;
        B.bg_tiles.clear();
    }
}

Segment acquire_segment()
{
    Ring &r = B.ring;
    if (REC.mode != 0)
    {
        // recording: the launches will be replayed, their descriptors have to stay -- a segment of its own, kept by the recorder
        char *h = nullptr, *d = nullptr, *twin = nullptr;
        HIP_CHECK(hipHostMalloc((void **)&h, r.seg_bytes, hipHostMallocNonCoherent | hipHostMallocMapped));
        HIP_CHECK(hipHostGetDevicePointer((void **)&d, h, 0));
        HIP_CHECK(hipMalloc((void **)&twin, r.seg_bytes));
        REC.segs.push_back(Recorder::Seg{h, d, twin, r.seg_bytes});
        REC.descriptor_bytes += r.seg_bytes;
        Segment s;
        s.h = h;
        s.d = d;
        s.cap = r.seg_bytes;
        s.used = 0;
        s.index = -1;
        return s;
    }
    int i = r.cur;
    r.cur = (r.cur + 1) % Ring::NSEG;
    if (r.used[i])
    {
        HostTimer ht(4);
        HIP_CHECK(hipEventSynchronize(r.ev[i])); // the kernels that last read this segment are done
    }
    Segment s;
    s.h = r.h + (size_t)i * r.seg_bytes;
    s.d = r.d + (size_t)i * r.seg_bytes;
    s.cap = r.seg_bytes;
    s.used = 0;
    s.index = i;
    return s;
}

// the segment is complete: kernels launched from now on may read it (in place, see ensure_ready)
void commit_segment(Segment &s)
{
    if (s.index >= 0)
        B.ring.pending.push_back(s.index);
}

hipEvent_t take_event()
{
    if (!B.event_pool.empty())
    {
        hipEvent_t e = B.event_pool.back();
        B.event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    HIP_CHECK(hipEventCreate(&e));
    return e;
}

struct LaunchTimer
{
    int cls;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    unsigned long long tag[3] = {0, 0, 0};
    explicit LaunchTimer(int c, hipStream_t stream = nullptr) : cls(c), st(stream ? stream : B.stream)
    {
        if (B.opt_profile)
        {
            a = take_event();
            b = take_event();
            HIP_CHECK(hipEventRecord(a, st));
        }
    }
    ~LaunchTimer()
    {
        if (B.opt_profile)
        {
            HIP_CHECK(hipEventRecord(b, st));
            B.pending_events.push_back(EventPair{a, b, cls, {tag[0], tag[1], tag[2]}});
        }
    }
};

void harvest_events()
{
    // PANGULU_HIP_LAUNCH_LOG=<file> (with PROFILE on): one line per launch -- class, microseconds, workgroups, tasks, live slab
    // steps -- for tuning the update kernel by launch shape (tools/launch_log_summary.py)
    static FILE *launch_log = getenv("PANGULU_HIP_LAUNCH_LOG") ? fopen(getenv("PANGULU_HIP_LAUNCH_LOG"), "w") : nullptr;
    for (auto &p : B.pending_events)
    {
        HIP_CHECK(hipEventSynchronize(p.b));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
        B.stats.elapsed_ms[p.cls] += ms;
        if (launch_log)
            fprintf(launch_log, "%d %.2f %llu %llu %llu\n", p.cls, 1e3 * ms, p.tag[0], p.tag[1], p.tag[2]);
        B.event_pool.push_back(p.a);
        B.event_pool.push_back(p.b);
    }
    B.pending_events.clear();
    if (launch_log)
        fflush(launch_log);
}

inline u32 host_nnz(const slot_t *s, int nb) { return s->columnpointer[nb]; }

// both halves of a diagonal block are one destination: name it by its lower half
inline slot_t *canon_dst(slot_t *s)
{
    if (s->brow_pos == s->bcol_pos && s->is_upper && s->related_block)
        return s->related_block;
    return s;
}

// identity of a block for the background-stream bookkeeping (both halves of a diagonal block are one block)
inline const void *block_key_any(const slot_t *s)
{
    if (s->brow_pos == s->bcol_pos && s->is_upper && s->related_block)
        return (const void *)s->related_block->d_value;
    return (const void *)s->d_value;
}

// Building the descriptors of a task touches, per operand, the slot struct, the last entry of its pattern pointer array and
// its block-table entry -- nine cache misses per update, and the leaf levels of the bench matrix (8000 updates + 4000
// solves per level) were bound by this thread, not by the device.  Two-stage software prefetch, a fixed distance ahead in
// the task list: the slot structs first, then what their fields point to.
inline void prefetch_task_slots(const task_t *t)
{
    for (const slot_t *s : {t->op1, t->op2, t->opdst})
        if (s)
        {
            __builtin_prefetch(s);
            __builtin_prefetch(reinterpret_cast<const char *>(s) + 64);
            __builtin_prefetch(reinterpret_cast<const char *>(s) + 128);
        }
}
void prefetch_task_details(const task_t *t, int nb); // (needs the block table: defined after pg_hip_dense_host.h)

inline void diag_halves(slot_t *any, slot_t **upper, slot_t **lower)
{
    if (any->is_upper)
    {
        *upper = any;
        *lower = any->related_block;
    }
    else
    {
        *upper = any->related_block;
        *lower = any;
    }
    if (!*upper || !*lower)
    {
        fprintf(stderr, "[PanguLU-AMD ERROR] diagonal block (%u,%u) is missing its other half\n", any->brow_pos, any->bcol_pos);
        exit(EXIT_FAILURE);
    }
}

// column view of the upper half of a diagonal block (needed when it is an SSSSM destination: the update runs
// column by column, the half is stored by rows)
const DiagAux &get_diag_aux(slot_t *upper, int nb)
{
    auto it = B.diag_aux.find((const void *)upper->d_value);
    u32 nnz = host_nnz(upper, nb);
    if (it != B.diag_aux.end() && it->second.nnz == nnz && it->second.brow == upper->brow_pos)
        return it->second;
    DiagAux aux;
    aux.nnz = nnz;
    aux.brow = upper->brow_pos;
    const u32 *rp = upper->columnpointer; // CSR row pointer (host naming, see pangulu_platform.h)
    const u16 *ci = upper->rowindex;
    std::vector<u32> cp(nb + 1, 0), vi(nnz);
    std::vector<u16> ri(nnz);
    for (u32 p = 0; p < nnz; p++)
        cp[ci[p] + 1]++;
    for (int c = 0; c < nb; c++)
        cp[c + 1] += cp[c];
    std::vector<u32> cur(cp.begin(), cp.end() - 1);
    for (int r = 0; r < nb; r++)
        for (u32 p = rp[r]; p < rp[r + 1]; p++)
        {
            u32 o = cur[ci[p]]++;
            ri[o] = (u16)r;
            vi[o] = p;
        }
    size_t bytes_cp = sizeof(u32) * (nb + 1), bytes_vi = sizeof(u32) * nnz, bytes_ri = sizeof(u16) * nnz;
    char *d = nullptr;
    size_t off_vi = (bytes_cp + 15) & ~(size_t)15, off_ri = (off_vi + bytes_vi + 15) & ~(size_t)15;
    HIP_CHECK(hipMalloc((void **)&d, off_ri + bytes_ri + 16));
    HIP_CHECK(hipMemcpy(d, cp.data(), bytes_cp, hipMemcpyHostToDevice));
    if (nnz)
    {
        HIP_CHECK(hipMemcpy(d + off_vi, vi.data(), bytes_vi, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(d + off_ri, ri.data(), bytes_ri, hipMemcpyHostToDevice));
    }
    aux.d_cp = (u32 *)d;
    aux.d_vi = (u32 *)(d + off_vi);
    aux.d_ri = (u16 *)(d + off_ri);
    if (it != B.diag_aux.end())
    {
        HIP_CHECK(hipFree(it->second.d_cp));
        it->second = aux;
        return it->second;
    }
    return B.diag_aux.emplace((const void *)upper->d_value, aux).first->second;
}

void mirror_to_host(slot_t *s, int nb)
{
    size_t bytes = sizeof(val_t) * (size_t)host_nnz(s, nb);
    join_records(B.stream);
    join_background(B.stream);
    if (bytes)
        HIP_CHECK(hipMemcpyAsync(s->value, s->d_value, bytes, hipMemcpyDeviceToHost, B.stream));
}

const double SV = (double)sizeof(val_t);

#include "pg_hip_dense_host.h"

void prefetch_task_details(const task_t *t, int nb)
{
    for (slot_t *s : {t->op1, t->op2, t->opdst})
        if (s)
        {
            if (s->columnpointer)
                __builtin_prefetch(&s->columnpointer[nb]);
#if defined(PG_DENSE_UPDATES)
            MP.blocks.prefetch(block_key(s));
#endif
        }
}
constexpr size_t PREFETCH_SLOTS_AHEAD = 24, PREFETCH_DETAILS_AHEAD = 12;

// ---- SSSSM -----------------------------------------------------------------------------------------------------
// Tasks arrive grouped by destination.  Per group the destination is either dense-mode (updates accumulate in its
// mirror) or sparse; per task the update runs on the matrix cores when destination and both operands have mirrors,
// on the LDS-accumulator kernel otherwise.
#define DG_TILE_HOST 128 // = DG_TILE of pg_hip_dense.h (R64 only; harmless elsewhere)
// Tasks per launch (PANGULU_HIP_LAUNCH_CHUNK).  The host builds the descriptors of a launch before it can start: a leaf level
// of the bench matrix has 8000 updates and 4000 solves, and the device sat idle for 260 us per level while their mirror
// jobs and descriptors were written.  Cut into chunks, the first kernels run while the rest is being prepared.
size_t launch_chunk_tasks()
{
    static const size_t chunk = []()
    {
        const char *e = getenv("PANGULU_HIP_LAUNCH_CHUNK");
        long v = e ? atol(e) : 0;
        return v > 0 ? (size_t)v : ~(size_t)0;
    }();
    return chunk;
}

// `background`: the update kernels of this call go to the background stream (see Backend::stream_bg); their mirror jobs
// stay on the main stream, in front of the fork
void launch_ssssm(int nb, task_t **list, size_t n, bool background = false)
{
    if (n == 0)
        return;
    HostTimer ht(0);
    hipStream_t const ms = background ? B.stream_bg : B.stream; // where the update kernels of this call run
    const bool dense_ok = dense_mode_available(nb);
    size_t i = 0;
    while (i < n)
    {
        Segment seg = acquire_segment();
        // worst case per task: one group + one task descriptor in each class; fill until the segment is full
        // (per update: a task descriptor in each class -- PG_PLANES^2 real products on the MFMA side --, a group in each, four
        // work items per MFMA group; the K-split of very small launches multiplies groups and work items of <= 64 tasks by four)
        const size_t per_task = sizeof(SsssmTaskD) * (1 + PG_PLANES * PG_PLANES) + sizeof(SsssmGroupD) * (1 + PG_PLANES) +
                                sizeof(SsssmWorkD) * 8 * PG_PLANES;
        size_t max_tasks = (seg.cap - 64 * 4 * PG_PLANES * (sizeof(SsssmGroupD) + 4 * sizeof(SsssmWorkD)) - 4096) / per_task;
        size_t take = std::min(n - i, std::min(max_tasks, launch_chunk_tasks()));
        SsssmTaskD *d_tasks_s, *d_tasks_d;
        SsssmGroupD *d_groups_s, *d_groups_d;
        SsssmTaskD *tasks_s = seg.alloc<SsssmTaskD>(take, &d_tasks_s);
        SsssmTaskD *tasks_d = seg.alloc<SsssmTaskD>(take * PG_PLANES * PG_PLANES, &d_tasks_d); // (CR64: four real products per update)
        const int tiles_per_dim = nb >= DG_TILE_HOST ? nb / DG_TILE_HOST : 1;
        const unsigned ksplit = (nb <= 256 && nb % 64 == 0 && take * (size_t)(tiles_per_dim * tiles_per_dim) <= 64) ? 4u : 1u;
        SsssmGroupD *groups_s = seg.alloc<SsssmGroupD>(take, &d_groups_s);
        SsssmGroupD *groups_d = seg.alloc<SsssmGroupD>(take * ksplit * PG_PLANES, &d_groups_d);
        static std::vector<unsigned short> live_k; // per dense task and tile: K-slabs in which both operands have entries
        live_k.assign(take * 4 * PG_PLANES * PG_PLANES, 0);
        static std::vector<unsigned char> full_t; // per dense task: tiles on which the update is a dense-front product
        full_t.assign(take * PG_PLANES * PG_PLANES, 0);
        SsssmWorkD *d_work, *d_work_f;
        SsssmWorkD *work = seg.alloc<SsssmWorkD>(take * ksplit * 4 * PG_PLANES, &d_work); // every workgroup of the MFMA launch ...
        SsssmWorkD *work_f = seg.alloc<SsssmWorkD>(take * 4 * PG_PLANES, &d_work_f);      // ... and of the dense-front launch
        if (!tasks_s || !tasks_d || !groups_s || !groups_d || !work || !work_f)
        {
            fprintf(stderr, "[PanguLU-AMD ERROR] descriptor staging segment too small\n");
            exit(EXIT_FAILURE);
        }
        size_t ns = 0, nd = 0, gs = 0, gd = 0, nd_updates = 0; // nd: real MFMA tasks; nd_updates: the updates they stand for
        double bytes_s = 0, bytes_d = 0;
        size_t end = i + take;
        while (i < end)
        {
            slot_t *dst = canon_dst(list[i]->opdst);
            size_t j = i;
            while (j < end && canon_dst(list[j]->opdst) == dst)
                j++;
            const bool diag = dst->brow_pos == dst->bcol_pos;
            SsssmGroupD G;
            memset(&G, 0, sizeof(G));
            u32 nnz_c;
            slot_t *up = nullptr, *lo = dst;
            if (diag)
            {
                diag_halves(dst, &up, &lo);
                nnz_c = host_nnz(lo, nb) + host_nnz(up, nb);
            }
            else
            {
                nnz_c = host_nnz(dst, nb);
            }
#if defined(PG_DENSE_UPDATES)
            // The destination works on its dense mirror when the mirror is already ahead of the sparse record, or
            // when at least one update of the group is heavy enough for the matrix cores.
            double *cm = nullptr;
            if (dense_ok)
            {
                bool want = mirror_is_ahead(dst);
                for (size_t t = i; t < j && !want; t++)
                    want = is_heavy_update(host_nnz(list[t]->op1, nb), host_nnz(list[t]->op2, nb), nb);
                if (want)
                    cm = current_mirror(dst, nb);
                if (cm)
                {
                    block_state(dst, nb).sparse_current = false; // from now on the mirror is ahead of the record
                    G.cdense = reinterpret_cast<val_t *>(cm);
                }
            }
            if (!cm)
                require_sparse(dst, nb);
#endif
            if (!G.cdense)
            {
                G.c = BlkView{lo->d_columnpointer, lo->d_rowindex, lo->d_value};
                if (diag)
                {
                    const DiagAux &aux = get_diag_aux(up, nb);
                    G.ucp = aux.d_cp;
                    G.uri = aux.d_ri;
                    G.uvi = aux.d_vi;
                    G.uval = up->d_value;
                }
            }
            size_t s0 = ns;
#if defined(PG_DENSE_UPDATES)
            // updates of this destination that go to the matrix cores: (operand mirrors, live K-slabs per tile)
            struct Heavy
            {
                SsssmTaskD T;
                unsigned short live[4];
                unsigned char full; // bit tl: every 16 x 16 piece of both operands that meets tile tl is live (dense front)
            };
            static thread_local std::vector<Heavy> heavy;
            heavy.clear();
#endif
            for (size_t t = i; t < j; t++)
            {
                if (t + PREFETCH_SLOTS_AHEAD < n)
                    prefetch_task_slots(list[t + PREFETCH_SLOTS_AHEAD]);
                if (t + PREFETCH_DETAILS_AHEAD < n)
                    prefetch_task_details(list[t + PREFETCH_DETAILS_AHEAD], nb);
                slot_t *a = list[t]->op1, *b = list[t]->op2;
                SsssmTaskD T;
                memset(&T, 0, sizeof(T));
                T.a = BlkView{a->d_columnpointer, a->d_rowindex, a->d_value};
                T.b = BlkView{b->d_columnpointer, b->d_rowindex, b->d_value};
                T.sign = 1.0;
                T.count = 1;
                u32 na = host_nnz(a, nb), nbz = host_nnz(b, nb);
                double by = (SV + 2) * ((double)na + nbz) + (2 * SV + 2) * (double)nnz_c + 12.0 * (nb + 1);
                bool on_mfma = false;
#if defined(PG_DENSE_UPDATES)
                if (G.cdense && is_heavy_update(na, nbz, nb))
                {
                    double *am = current_mirror(a, nb);
                    double *bm = am ? current_mirror(b, nb) : nullptr;
                    if (am && bm)
                    {
                        T.a.val = reinterpret_cast<val_t *>(am); // the pattern pointers stay: the flop counter reads them
                        T.b.val = reinterpret_cast<val_t *>(bm);
                        on_mfma = true;
                    }
                }
                if (on_mfma)
                {
                    // tiles of the destination this update can reach (tile = tm + tiles * tn), per K-slab
                    Heavy H;
                    H.T = T;
                    const BlockState *sa = MP.blocks.find(block_key(a)), *sb = MP.blocks.find(block_key(b));
                    for (int tl = 0; tl < 4; tl++)
                        H.live[tl] = tl >= tiles_per_dim * tiles_per_dim ? (unsigned short)0
                                     : (sa && sb && sa->occ_valid && sb->occ_valid)
                                         ? (unsigned short)(sa->occ_a[tl % tiles_per_dim] & sb->occ_b[tl / tiles_per_dim])
                                         : (unsigned short)0xFFFF;
                    H.full = 0;
                    if (sa && sb && sa->occ_valid && sb->occ_valid)
                    {
                        H.T.has_map = 1;
                        memcpy(H.T.amap, sa->occ_map, sizeof(H.T.amap));
                        memcpy(H.T.bmap_t, sb->occ_map_t, sizeof(H.T.bmap_t));
                        const int nslab = nb / 16;
                        const unsigned pm = nb >= 128 ? 0xFFu : ((1u << nslab) - 1u);
                        for (int tl = 0; tl < tiles_per_dim * tiles_per_dim; tl++)
                        {
                            const int tm = tl % tiles_per_dim, tn = tl / tiles_per_dim;
                            bool all = true;
                            for (int sl = 0; sl < nslab && all; sl++)
                                all = (((unsigned)H.T.amap[sl] >> (8 * tm)) & pm) == pm && (((unsigned)H.T.bmap_t[sl] >> (8 * tn)) & pm) == pm;
                            if (all)
                                H.full |= (unsigned char)(1u << tl);
                        }
                    }
                    heavy.push_back(H);
                    bytes_d += by;
                    nd_updates++;
                }
#endif
                if (!on_mfma)
                {
                    tasks_s[ns++] = T;
                    bytes_s += by;
                }
            }
            // cut long queues into chunks that run concurrently and merge with atomics
            // (a launch with few updates cannot fill the chip with whole queues: one update per workgroup then)
            size_t chunk = (size_t)(B.opt_group_chunk > 0 ? B.opt_group_chunk : 1 << 30);
            if (B.opt_group_chunk > 0 && take <= (size_t)B.opt_small_launch_tasks)
                chunk = 1;
            size_t nheavy = 0;
#if defined(PG_DENSE_UPDATES)
            nheavy = heavy.size();
#endif
            // ... and a destination updated by both kernels at once (they run side by side on two streams) must take
            // atomics from both
            const bool split = (ns - s0) > chunk || nheavy > chunk || ((ns > s0) && nheavy && B.opt_two_streams);
            for (size_t c = s0; c < ns; c += chunk)
            {
                G.task_begin = (u32)c;
                G.task_end = (u32)std::min(ns, c + chunk);
                G.atomic = split ? 1u : 0u;
                groups_s[gs++] = G;
            }
#if defined(PG_DENSE_UPDATES)
            // R64: one task per update.  CR64: per destination plane the two real products of every update, consecutive, so
            // that one accumulator pass serves both (C_re -= A_re B_re - A_im B_im;  C_im -= A_re B_im + A_im B_re)
            for (int plane = 0; plane < PG_PLANES && nheavy; plane++)
            {
                const size_t d0 = nd;
                for (const Heavy &H : heavy)
                    for (int term = 0; term < PG_PLANES; term++)
                    {
                        SsssmTaskD T = H.T;
#if PG_PLANES > 1
                        const size_t ps = mirror_plane_stride(nb);
                        double *am = reinterpret_cast<double *>(H.T.a.val), *bm = reinterpret_cast<double *>(H.T.b.val);
                        // plane 0 (real):  + A_re B_re  - A_im B_im      plane 1 (imaginary):  + A_re B_im  + A_im B_re
                        const int a_im = term, b_im = plane ^ term;
                        T.a.val = reinterpret_cast<val_t *>(am + (a_im ? ps : 0));
                        T.b.val = reinterpret_cast<val_t *>(bm + (b_im ? ps : 0));
                        T.sign = (plane == 0 && term == 1) ? -1.0 : 1.0;
                        T.count = (plane == 0 && term == 0) ? 1u : 0u;
#endif
                        for (int tl = 0; tl < 4; tl++)
                            live_k[nd * 4 + tl] = H.live[tl];
                        full_t[nd] = H.full;
                        tasks_d[nd++] = T;
                    }
                SsssmGroupD GP = G;
#if PG_PLANES > 1
                GP.cdense = reinterpret_cast<val_t *>(reinterpret_cast<double *>(G.cdense) + (size_t)plane * mirror_plane_stride(nb));
#endif
                const size_t dchunk = chunk >= ((size_t)1 << 28) ? chunk : chunk * PG_PLANES;
                for (size_t c = d0; c < nd; c += dchunk)
                {
                    GP.task_begin = (u32)c;
                    GP.task_end = (u32)std::min(nd, c + dchunk);
                    GP.atomic = (split || ksplit > 1) ? 1u : 0u;
                    const unsigned slabs = (unsigned)nb / 16u, per = slabs / ksplit;
                    for (unsigned q = 0; q < ksplit; q++)
                    {
                        GP.slab_mask = ksplit > 1 ? (((1u << per) - 1u) << (q * per)) : 0u;
                        const unsigned kmask = GP.slab_mask ? GP.slab_mask : 0xFFFFu;
                        GP.live_tiles = 0;
                        for (u32 t = GP.task_begin; t < GP.task_end; t++)
                            for (int tl = 0; tl < tiles_per_dim * tiles_per_dim; tl++)
                                if (live_k[(size_t)t * 4 + tl] & kmask)
                                    GP.live_tiles |= 1u << tl;
                        groups_d[gd++] = GP;
                    }
                }
            }
#endif
            G.slab_mask = 0;
            G.live_tiles = 0;
            i = j;
        }
#if defined(PG_DENSE_UPDATES)
        // mirrors that have to be (re)built for this launch, and sparse records that must catch up first
        if (!MP.to_sparsify.empty())
            flush_mirror_jobs(nb, MP.to_sparsify, false);
        if (!MP.to_densify.empty())
            flush_mirror_jobs(nb, MP.to_densify, true);
#endif
        // longest queues first: workgroups are dispatched in grid order, so the big groups start at once and the small
        // ones fill the tail of the launch
        auto by_size = [](const SsssmGroupD &x, const SsssmGroupD &y)
        { return (x.task_end - x.task_begin) > (y.task_end - y.task_begin); };
        std::stable_sort(groups_s, groups_s + gs, by_size);
        std::stable_sort(groups_d, groups_d + gd, by_size);
        commit_segment(seg);
        if (background)
        {
            // mirrors are current and the operands final from here on (main stream); the kernels run on the background stream
            pg_event_record(B.ev_bg_fork, B.stream);
            pg_stream_wait(ms, B.ev_bg_fork);
        }
        if (gs && gd && B.opt_two_streams && !background)
            pg_event_record(B.ev_fork, B.stream); // mirrors are current from here on
        if (gs)
        {
            join_records(ms); // operands and destinations of the LDS kernel are sparse records
            LaunchTimer lt(4, ms);
            // columns per wavefront: 1 unless the grid would exceed 2^20 workgroups (more parallel waves beat fewer launches:
            // measured 176 ms vs 181 ms per factorisation of the bench matrix with an 8k-workgroup target)
            int cpw = 1;
            while (cpw < 64 && gs * (size_t)((nb + SSSSM_WAVES * cpw - 1) / (SSSSM_WAVES * cpw)) > ((size_t)1 << 20))
                cpw *= 2;
            int colblocks = (nb + SSSSM_WAVES * cpw - 1) / (SSSSM_WAVES * cpw);
            size_t lds = sizeof(val_t) * (size_t)nb * SSSSM_WAVES;
            if (B.opt_getrf_strict)
                PG_LAUNCH(ssssm_sparse_kernel<true>, dim3((unsigned)(gs * colblocks)), dim3(SSSSM_WAVES * 64), lds, ms,
                                   d_groups_s, d_tasks_s, nb, cpw, B.d_flops + 4);
            else
                PG_LAUNCH(ssssm_sparse_kernel<false>, dim3((unsigned)(gs * colblocks)), dim3(SSSSM_WAVES * 64), lds, ms,
                                   d_groups_s, d_tasks_s, nb, cpw, B.d_flops + 4);
            B.stats.launches[4]++;
            B.stats.tasks[4] += ns;
            B.stats.alg_bytes[4] += bytes_s;
        }
#if defined(PG_DENSE_UPDATES)
        if (gd)
        {
            hipStream_t ds = ms;
            const bool side = B.opt_two_streams && gs && !background;
            if (side)
            {
                // fork: the MFMA kernel starts as soon as the mirrors are ready and runs beside the LDS kernel (both
                // are bound by memory latency and launch tails, not by a shared resource)
                ds = B.stream2;
                pg_stream_wait(ds, B.ev_fork);
            }
            {
                // one workgroup per (group, tile) some update of the group can reach.  Pairs whose whole queue is dense-front
                // products (every 16 x 16 piece of every operand live, no K-split) go to the front kernel's list
                int tiles = nb / DG_TILE;
                size_t nw = 0, nf = 0, nfm = 0;
                const bool front_on = B.opt_front_stages >= 1 && (nb == 128 || nb == 256) && (B.opt_front_stages >= 2 || B.opt_tiles_stages >= 1);
                // (first pass: which pairs qualify, and how many -- a front launch of its own pays from a few thousand workgroups
                //  on: fem27(112) 883.8 ms with it against 892.1 with the pairs inside the general launch, shell(398) 39.2 against 38.5)
                static std::vector<unsigned char> full_g;
                full_g.assign(gd, 0);
                size_t nfull = 0;
                for (size_t gi = 0; gi < gd && front_on; gi++)
                {
                    const SsssmGroupD &Gd = groups_d[gi];
                    unsigned all_full = Gd.slab_mask ? 0u : 0xFu;
                    for (u32 t = Gd.task_begin; t < Gd.task_end && all_full; t++)
                        all_full &= full_t[t];
                    all_full &= Gd.live_tiles;
                    full_g[gi] = (unsigned char)all_full;
                    nfull += (size_t)__builtin_popcount(all_full);
                }
                const bool own_launch = B.opt_front_stages >= 2 && (B.opt_tiles_stages < 1 || nfull >= (size_t)B.opt_front_min_wgs);
                // Longest queues first (PANGULU_HIP_HEAVY_FIRST): a launch ends with its last workgroup, and a queue of 128 live slab
                // steps that starts when the others are done is a tail of its own length.  Classes by the live steps of a group's
                // busiest tile -- sixteen of 16 steps each (2, the default), or four (1) --, the scheduler's order kept inside a class
                // (neighbours share operands: L2).  fem27(112), one box: 810.3-811.6 / 813.3 / 815.4 ms with 2 / 1 / 0
                // (profiles/r03ak_heavy_first.log); shell(398) indifferent.
                static const int heavy_mode = getenv("PANGULU_HIP_HEAVY_FIRST") ? atoi(getenv("PANGULU_HIP_HEAVY_FIRST")) : 2;
                static const bool heavy_first = heavy_mode != 0;
                static std::vector<u32> g_order;
                g_order.resize(gd);
                if (heavy_first && gd > 1)
                {
                    static std::vector<unsigned char> g_class;
                    g_class.resize(gd);
                    size_t count[16] = {0};
                    for (size_t gi = 0; gi < gd; gi++)
                    {
                        const SsssmGroupD &Gd = groups_d[gi];
                        const unsigned kmask = Gd.slab_mask ? Gd.slab_mask : 0xFFFFu;
                        unsigned steps[4] = {0, 0, 0, 0};
                        for (u32 t = Gd.task_begin; t < Gd.task_end; t++)
                            for (int tl = 0; tl < tiles * tiles; tl++)
                                steps[tl] += (unsigned)__builtin_popcount(live_k[(size_t)t * 4 + tl] & kmask);
                        const unsigned most = std::max(std::max(steps[0], steps[1]), std::max(steps[2], steps[3]));
                        if (heavy_mode == 2)
                            g_class[gi] = (unsigned char)(15 - std::min(15u, most / 16u));
                        else
                            g_class[gi] = most >= 96 ? 0 : most >= 48 ? 1 : most >= 24 ? 2 : 3;
                        count[g_class[gi]]++;
                    }
                    size_t at[16];
                    at[0] = 0;
                    for (int c = 1; c < 16; c++)
                        at[c] = at[c - 1] + count[c - 1];
                    for (size_t gi = 0; gi < gd; gi++)
                        g_order[at[g_class[gi]]++] = (u32)gi;
                }
                else
                    for (size_t gi = 0; gi < gd; gi++)
                        g_order[gi] = (u32)gi;
                for (size_t go = 0; go < gd; go++)
                {
                    const size_t gi = g_order[go];
                    const SsssmGroupD &Gd = groups_d[gi];
                    const unsigned all_full = full_g[gi];
                    for (int tl = 0; tl < tiles * tiles; tl++)
                        if ((Gd.live_tiles >> tl) & 1u)
                        {
                            SsssmWorkD item{Gd.cdense, Gd.task_begin, Gd.task_end, Gd.atomic, Gd.slab_mask, (u32)tl, 0u};
                            if (!((all_full >> tl) & 1u))
                                work[nw++] = item;
                            else if (own_launch)
                                work_f[nf++] = item;
                            else
                            {
                                // same launch as the partly filled tiles: one launch, one tail; the kernel skips the step list
                                item.pad_ = 1u;
                                work[nw++] = item;
                                nfm++;
                            }
                        }
                }
                LaunchTimer lt(5, ds);
                if (B.opt_profile)
                {
                    unsigned long long steps = 0;
                    for (size_t gi = 0; gi < gd; gi++)
                    {
                        const SsssmGroupD &Gd = groups_d[gi];
                        const unsigned kmask = Gd.slab_mask ? Gd.slab_mask : 0xFFFFu;
                        for (u32 t = Gd.task_begin; t < Gd.task_end; t++)
                            for (int tl = 0; tl < tiles * tiles; tl++)
                                steps += (unsigned long long)__builtin_popcount(live_k[(size_t)t * 4 + tl] & kmask);
                    }
                    lt.tag[0] = nw + nf;
                    lt.tag[1] = nd;
                    lt.tag[2] = steps;
                }
                B.front_workgroups += nf + nfm;
                B.general_workgroups += nw - nfm;
                static const bool debug_ssssm = getenv("PANGULU_HIP_DEBUG_SSSSM") != nullptr; // (stamps share the GETRF debug slots)
                unsigned long long *pc = B.opt_count_flops ? B.d_flops + 6 : nullptr;
                if (nf)
                {
                    // the longest-running workgroups first: the front launch, then the general one fills in behind it
                    const unsigned unit = (unsigned)(tiles * tiles) * (unsigned)std::max<long long>(1, B.opt_front_unit);
                    if (B.opt_front_stages >= 4)
                        PG_LAUNCH((ssssm_front_f64_kernel<4, true>), dim3((unsigned)nf), dim3(FR_THREADS), 0, ds, d_tasks_d, nb, d_work_f, pc, unit);
                    else if (B.opt_front_stages == 3)
                        PG_LAUNCH((ssssm_front_f64_kernel<3, true>), dim3((unsigned)nf), dim3(FR_THREADS), 0, ds, d_tasks_d, nb, d_work_f, pc, unit);
                    else
                        PG_LAUNCH((ssssm_front_f64_kernel<2, true>), dim3((unsigned)nf), dim3(FR_THREADS), 0, ds, d_tasks_d, nb, d_work_f, pc, unit);
                }
                if (nw && B.opt_tiles_stages >= 2)
                {
                    // round 3: LDS-DMA pipeline, strided piece ownership (pg_hip_front.h)
                    const unsigned unit = (unsigned)(tiles * tiles) * (unsigned)std::max<long long>(1, B.opt_tiles_unit);
                    if (B.opt_tiles_stages >= 4)
                        PG_LAUNCH((ssssm_tiles_f64_kernel<4>), dim3((unsigned)nw), dim3(FR_THREADS), 0, ds, d_tasks_d, nb, d_work, pc, unit);
                    else if (B.opt_tiles_stages == 3)
                        PG_LAUNCH((ssssm_tiles_f64_kernel<3>), dim3((unsigned)nw), dim3(FR_THREADS), 0, ds, d_tasks_d, nb, d_work, pc, unit);
                    else if (B.opt_tiles_stages == 2)
                        // (the default: two stages, step records prefetched, DMA issue behind the first products)
                        PG_LAUNCH(ssssm_tilesv_f64_kernel, dim3((unsigned)nw), dim3(FR_THREADS), 0, ds, d_tasks_d, nb, d_work, pc, unit);
                }
                else if (nw && B.opt_tiles_stages == 1)
                {
                    // (the first two-stage version: DMA issue right behind the barrier)
                    const unsigned unit = (unsigned)(tiles * tiles) * (unsigned)std::max<long long>(1, B.opt_tiles_unit);
                    PG_LAUNCH((ssssm_tiles_f64_kernel<2>), dim3((unsigned)nw), dim3(FR_THREADS), 0, ds, d_tasks_d, nb, d_work, pc, unit);
                }
                else if (nw)
                    PG_LAUNCH(ssssm_dense_f64_kernel, dim3((unsigned)nw), dim3(DG_THREADS), 0, ds, d_tasks_d, nb, pc,
                                       debug_ssssm ? B.d_flops + 8 : nullptr, d_work);
            }
            if (B.opt_count_flops)
                PG_LAUNCH(ssssm_flop_count_kernel, dim3((unsigned)nd), dim3(256), 0, ds, d_tasks_d, nb, B.d_flops + 5);
            if (side)
            {
                pg_event_record(B.ev_join, ds);
                pg_stream_wait(B.stream, B.ev_join); // join before anything later on the main stream
            }
            B.stats.launches[5]++;
            B.stats.tasks[5] += nd_updates;
            B.stats.alg_bytes[5] += bytes_d;
        }
#endif
        HIP_CHECK(hipGetLastError());
        release_pending_segments(ms);
    }
    if (background)
    {
        pg_event_record(B.ev_bg_done, ms);
        B.bg_active = true;
        for (size_t t = 0; t < n; t++)
            B.bg_tiles.insert(block_key_any(list[t]->opdst));
    }
}

// ---- TSTRF / GESSM -----------------------------------------------------------------------------------------------
void launch_trsm(int nb, task_t **list, size_t n)
{
    HostTimer ht(1);
    size_t i = 0;
    PEND.hold = PEND.active; // (a held factorisation waits until this call knows whether its solves can chase it)
    while (i < n)
    {
        Segment seg = acquire_segment();
        size_t take = std::min(n - i, seg.cap / (sizeof(TrsmTaskD) + sizeof(TrsmTaskD) + 64 + 80 + 4 * sizeof(u32))); // (+80: a remote-diagonal image job per task at worst)
        take = std::min(take, launch_chunk_tasks());
        {
            // PANGULU_HIP_TRSM_CHUNK: solves per launch (0 = all).  The leaf levels of a large problem bring tens of thousands of
            // solves in one call, and the device sits empty while their mirror jobs and descriptors are written
            static const size_t trsm_chunk = []()
            {
                const char *e = getenv("PANGULU_HIP_TRSM_CHUNK");
                const long v = e ? atol(e) : 0;
                return v > 0 ? (size_t)v : ~(size_t)0;
            }();
            take = std::min(take, trsm_chunk);
        }
        TrsmTaskD *d_tasks, *d_ftasks;
        TrsmTaskD *tasks = seg.alloc<TrsmTaskD>(take, &d_tasks);
        TrsmTaskD *ftasks = seg.alloc<TrsmTaskD>(take, &d_ftasks); // sparse views of the dense-path tasks (flop counting)
        double by_t = 0, by_g = 0;
        size_t nt = 0, ng = 0, nsparse = 0, ndense = 0;
#if defined(PG_DENSE_PANELS)
        TrsmDenseTaskD *d_dtasks;
        TrsmDenseTaskD *dtasks = seg.alloc<TrsmDenseTaskD>(take, &d_dtasks);
        u32 *d_dwork;
        u32 *dwork = seg.alloc<u32>(take * 4, &d_dwork); // (task, 64-wide slab) of every workgroup of the dense-solve launch
        static std::vector<unsigned short> dlive;        // per dense task: which 16-wide strips of the block hold entries
        dlive.assign(take, 0);
        std::vector<slot_t *> solved_dense;
        const bool dense_ok = dense_mode_available(nb);
#endif
#if defined(PG_COMPLEX_PANELS)
        static const bool zpanels_on = !(getenv("PANGULU_HIP_COMPLEX_PANELS") && atoi(getenv("PANGULU_HIP_COMPLEX_PANELS")) == 0);
        std::vector<ZTrsmTaskD> zt; // (block, 64-wide slab) items of the solves that run on mirrors (ztrsm_planes_kernel)
        std::vector<slot_t *> solved_dense;
        const bool dense_ok = dense_mode_available(nb);
#endif
        for (size_t k = 0; k < take; k++)
        {
            if (i + k + PREFETCH_SLOTS_AHEAD < n)
                prefetch_task_slots(list[i + k + PREFETCH_SLOTS_AHEAD]);
            if (i + k + PREFETCH_DETAILS_AHEAD < n)
                prefetch_task_details(list[i + k + PREFETCH_DETAILS_AHEAD], nb);
            task_t *t = list[i + k];
            slot_t *dst = t->opdst, *diag = t->op1;
            // opdiag may be either half (…0100000.c:143-145,184-186); only the half the solve reads has to exist
            // (a rank that received a remote diagonal for its TSTRFs only may never get the L half)
            const bool want_upper = t->kernel_id == PANGULU_TASK_TSTRF;
            slot_t *half = ((diag->is_upper != 0) == want_upper) ? diag : diag->related_block;
            if (!half)
            {
                fprintf(stderr, "[PanguLU-AMD ERROR] %s on block (%u,%u): the %s half of diagonal %u is not available\n",
                        want_upper ? "TSTRF" : "GESSM", dst->brow_pos, dst->bcol_pos, want_upper ? "upper" : "lower", diag->brow_pos);
                exit(EXIT_FAILURE);
            }
            slot_t *up = half, *lo = half;
            TrsmTaskD T;
            memset(&T, 0, sizeof(T));
            u32 nnz_b = host_nnz(dst, nb);
            if (t->kernel_id == PANGULU_TASK_TSTRF)
            {
                T.vptr = dst->d_rowpointer;
                T.vidx = dst->d_columnindex;
                T.vmap = dst->d_idx_of_csc_value_for_csr;
                T.bval = dst->d_value;
                T.tptr = up->d_rowpointer;
                T.tidx = up->d_columnindex;
                T.tval = up->d_value;
                T.is_tstrf = 1;
                by_t += (2 * SV + 6) * (double)nnz_b + 4.0 * (nb + 1) + (SV + 2) * (double)host_nnz(up, nb) + 4.0 * (nb + 1);
                nt++;
            }
            else
            {
                T.vptr = dst->d_columnpointer;
                T.vidx = dst->d_rowindex;
                T.vmap = nullptr;
                T.bval = dst->d_value;
                T.tptr = lo->d_columnpointer;
                T.tidx = lo->d_rowindex;
                T.tval = lo->d_value;
                T.is_tstrf = 0;
                by_g += (2 * SV + 2) * (double)nnz_b + 4.0 * (nb + 1) + (SV + 2) * (double)host_nnz(lo, nb) + 4.0 * (nb + 1);
                ng++;
            }
            bool dense = false;
#if defined(PG_DENSE_PANELS)
            // dense path: the diagonal block left a dense LU image with inverted diagonal tiles (launch_getrf) and the
            // block being solved is well filled or already lives in its mirror
            if (dense_ok && (nb == 128 || nb == 256) && B.opt_trsm_dense_permille <= 1000)
            {
                const double *lu = lu_image_of(half);
                const bool filled = (u64)nnz_b * 1000ull >= (u64)B.opt_trsm_dense_permille * (u64)nb * (u64)nb;
                if (!lu && (filled || mirror_is_ahead(dst)))
                    lu = request_half_image(half, nb); // a diagonal block another rank factorised
                if (lu && (filled || mirror_is_ahead(dst)))
                {
                    double *bm = current_mirror(dst, nb);
                    if (bm)
                    {
                        TrsmDenseTaskD D;
                        D.b = bm;
                        D.lu = lu;
                        D.is_tstrf = T.is_tstrf;
                        D.lu_map = lu_image_has_map(half) ? 1u : 0u;
                        {
                            // strips of the solve = row slabs (TSTRF) / column slabs (GESSM) of the block
                            const BlockState *sd = MP.blocks.find(block_key(dst));
                            dlive[ndense] = (sd && sd->occ_valid) ? (T.is_tstrf ? sd->occ_rows : sd->occ_cols) : (unsigned short)0xFFFF;
                        }
                        dtasks[ndense] = D;
                        // the flop counter wants the CSC view of the block in both cases
                        T.vptr = dst->d_columnpointer;
                        T.vidx = dst->d_rowindex;
                        ftasks[ndense++] = T;
                        solved_dense.push_back(dst);
                        dense = true;
                    }
                }
            }
#endif
#if defined(PG_COMPLEX_PANELS)
            // complex types: the diagonal block was factorised in its mirror (launch_getrf) and the block being solved is well filled
            // or already lives in its mirror: solve it there, one workgroup per 64-wide slab that holds pattern entries
            if (zpanels_on && dense_ok && (nb == 128 || nb == 256) && B.opt_trsm_dense_permille <= 1000 && !B.opt_host_mirror)
            {
                const double *lu = lu_image_of(half);
                const bool filled = (u64)nnz_b * 1000ull >= (u64)B.opt_trsm_dense_permille * (u64)nb * (u64)nb;
                if (lu && (filled || mirror_is_ahead(dst)))
                {
                    double *bm = current_mirror(dst, nb);
                    if (bm)
                    {
                        const BlockState *sd = MP.blocks.find(block_key(dst));
                        const unsigned live = (sd && sd->occ_valid) ? (T.is_tstrf ? sd->occ_rows : sd->occ_cols) : 0xFFFFu;
                        for (int w = 0; w < nb / 64; w++)
                            if ((live >> (4 * w)) & 0xFu)
                                zt.push_back(ZTrsmTaskD{bm, lu, (u32)T.is_tstrf, (u32)w});
                        // the flop counter wants the CSC view of the block in both cases
                        T.vptr = dst->d_columnpointer;
                        T.vidx = dst->d_rowindex;
                        ftasks[ndense++] = T;
                        solved_dense.push_back(dst);
                        dense = true;
                    }
                }
            }
#endif
            if (!dense)
            {
                require_sparse(dst, nb); // updates may have been accumulating in the block's mirror
                tasks[nsparse++] = T;
#if defined(PG_DENSE_UPDATES)
                if (BlockState *found = MP.blocks.find(block_key(dst)))
                    found->mirror_current = false; // the sparse solve rewrites the record
#endif
            }
        }
#if defined(PG_DENSE_UPDATES)
        if (!MP.to_sparsify.empty())
            flush_mirror_jobs(nb, MP.to_sparsify, false);
        if (!MP.to_densify.empty())
            flush_mirror_jobs(nb, MP.to_densify, true);
#endif
#if defined(PG_DENSE_PANELS)
        if (!g_half_image_jobs.empty())
        {
            // images of remote diagonal blocks: build, then invert their diagonal tiles (main stream, before the solves)
            const size_t nj = g_half_image_jobs.size();
            HalfImageJobD *d_jobs;
            HalfImageJobD *hj = seg.alloc<HalfImageJobD>(nj, &d_jobs);
            double **d_imgs;
            double **imgs = seg.alloc<double *>(nj, &d_imgs);
            if (!hj || !imgs)
            {
                fprintf(stderr, "[PanguLU-AMD ERROR] descriptor staging segment too small\n");
                exit(EXIT_FAILURE);
            }
            for (size_t q = 0; q < nj; q++)
            {
                hj[q] = g_half_image_jobs[q];
                imgs[q] = g_half_image_jobs[q].dense;
            }
            g_half_image_jobs.clear();
            {
                LaunchTimer lt(8);
                PG_LAUNCH(half_image_kernel, dim3((unsigned)nj), dim3(1024), sizeof(u32) * (size_t)(nb + 1), B.stream, d_jobs, nb);
                PG_LAUNCH(diag_tile_inverse_kernel, dim3((unsigned)(nj * (nb / 16))), dim3(64), 0, B.stream, d_imgs, nb);
            }
            B.stats.launches[8]++;
            B.stats.tasks[8] += nj;
            B.stats.alg_bytes[8] += (double)nj * sizeof(double) * nb * nb;
            HIP_CHECK(hipGetLastError());
        }
#endif
#if defined(PG_COMPLEX_PANELS)
        ZTrsmTaskD *d_zt = nullptr;
        if (!zt.empty())
        {
            ZTrsmTaskD *hz = seg.alloc<ZTrsmTaskD>(zt.size(), &d_zt);
            if (!hz)
            {
                fprintf(stderr, "[PanguLU-AMD ERROR] descriptor staging segment too small\n");
                exit(EXIT_FAILURE);
            }
            memcpy(hz, zt.data(), sizeof(ZTrsmTaskD) * zt.size());
        }
#endif
        commit_segment(seg);
#if defined(PG_DENSE_PANELS)
        // chase: every solve of this call is a dense one against an image the held factorisation is going to leave
        static const bool direct_solves = getenv("PANGULU_HIP_TRSM_DIRECT") ? atoi(getenv("PANGULU_HIP_TRSM_DIRECT")) != 0 : true;
        bool chase = PEND.active && PEND.hold && i == 0 && take == n && ndense > 0 && nsparse == 0 && direct_solves && PEND.nb == nb;
        for (size_t t = 0; t < ndense && chase; t++)
        {
            size_t at = 0;
            while (at < PEND.images.size() && PEND.images[at] != dtasks[t].lu)
                at++;
            chase = at < PEND.images.size();
            if (chase)
                dtasks[t].progress = PEND.d_progress + at;
        }
        PEND.hold = false;
        if (!chase)
            flush_pending_getrf(); // (as it was: the factorisation, then this call's kernels)
        if (ndense && nsparse && B.opt_two_streams)
            pg_event_record(B.ev_fork, B.stream); // mirrors and sparse records are current from here on
#endif
        {
            LaunchTimer lt(nt >= ng ? 2 : 3);
            if (nsparse)
            {
                join_records(B.stream); // the sparse solves read the diagonal halves' records (behind the fork: the dense solves do not wait)
                int vblocks = (nb + TRSM_WAVES - 1) / TRSM_WAVES;
                size_t lds = sizeof(val_t) * (size_t)nb * TRSM_WAVES;
                PG_LAUNCH(trsm_sparse_kernel, dim3((unsigned)(nsparse * vblocks)), dim3(TRSM_WAVES * 64), lds, B.stream, d_tasks,
                                   nb, B.d_flops + 2, B.d_flops + 3);
            }
#if defined(PG_COMPLEX_PANELS)
            if (!zt.empty())
            {
                const size_t lds_z = sizeof(double) * 2 * ZP_PANEL * (size_t)nb;
                static size_t zt_allowed = 0;
                if (lds_z > zt_allowed)
                {
                    HIP_CHECK(hipFuncSetAttribute((const void *)ztrsm_planes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_z));
                    zt_allowed = lds_z;
                }
                PG_LAUNCH(ztrsm_planes_kernel, dim3((unsigned)zt.size()), dim3(ZT_THREADS), lds_z, B.stream, (const ZTrsmTaskD *)d_zt, nb);
            }
#endif
#if defined(PG_DENSE_PANELS)
            if (ndense)
            {
                // the dense solves run beside the sparse ones (other blocks, same diagonal operands)
                hipStream_t ds = (B.opt_two_streams && nsparse) ? B.stream2 : B.stream;
                if (ds != B.stream)
                    pg_stream_wait(ds, B.ev_fork);
                static const bool debug_trsm = getenv("PANGULU_HIP_DEBUG_TRSM") != nullptr; // (stamps share the GETRF debug slots)
                // barrier-free kernel by default (PANGULU_HIP_TRSM_DIRECT=0: the LDS-staged one)
                static const bool direct = getenv("PANGULU_HIP_TRSM_DIRECT") ? atoi(getenv("PANGULU_HIP_TRSM_DIRECT")) != 0 : true;
                unsigned long long *dbg = debug_trsm ? B.d_flops + 8 : nullptr;
                // one workgroup per (task, 64-wide slab) that holds pattern entries
                size_t nw = 0;
                for (size_t t = 0; t < ndense; t++)
                    for (int w = 0; w < nb / 64; w++)
                        if ((dlive[t] >> (4 * w)) & 0xFu)
                            dwork[nw++] = (u32)(t << 2) | (u32)w;
                if (chase)
                {
                    // one launch: the held factorisation's workgroups first, then two (task, slab) items per workgroup
                    const size_t lds_t = gt_lds_bytes(nb);
                    static size_t c_allowed = 0;
                    if (lds_t > c_allowed)
                    {
                        HIP_CHECK(hipFuncSetAttribute((const void *)getrf_trsm_chase_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
                        HIP_CHECK(hipFuncSetAttribute((const void *)getrf_trsm_chase_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
                        c_allowed = lds_t;
                    }
                    PendingGetrf P = std::move(PEND);
                    PEND = PendingGetrf();
                    const unsigned ng_ = (unsigned)P.take, nwg = ng_ + (unsigned)((nw + 1) / 2);
                    const GetrfTaskD *gt_ = static_cast<const GetrfTaskD *>(P.d_tasks);
                    PG_LAUNCH(zero_words_kernel, dim3(1), dim3(256), 0, ds, P.d_progress, ng_);
                    if (nb == 256)
                        PG_LAUNCH(getrf_trsm_chase_kernel<16>, dim3(nwg), dim3(GT_THREADS), lds_t, ds, gt_, ng_, P.d_progress, B.d_flops + 1, (const TrsmDenseTaskD *)d_dtasks,
                                  (const u32 *)d_dwork, (unsigned)nw);
                    else
                        PG_LAUNCH(getrf_trsm_chase_kernel<8>, dim3(nwg), dim3(GT_THREADS), lds_t, ds, gt_, ng_, P.d_progress, B.d_flops + 1, (const TrsmDenseTaskD *)d_dtasks,
                                  (const u32 *)d_dwork, (unsigned)nw);
                    P.post();
                    B.chase_launches++;
                    B.chase_solves += ndense;
                }
                else if (!nw)
                    ;
                else if (direct && nb == 256)
                    PG_LAUNCH(trsm_dense_direct_f64_kernel<16>, dim3((unsigned)nw), dim3(256), 0, ds, d_dtasks, d_dwork);
                else if (direct)
                    PG_LAUNCH(trsm_dense_direct_f64_kernel<8>, dim3((unsigned)nw), dim3(256), 0, ds, d_dtasks, d_dwork);
                else if (nb == 256)
                    PG_LAUNCH(trsm_dense_f64_kernel<16>, dim3((unsigned)nw), dim3(256), 0, ds, d_dtasks, dbg, d_dwork);
                else
                    PG_LAUNCH(trsm_dense_f64_kernel<8>, dim3((unsigned)nw), dim3(256), 0, ds, d_dtasks, dbg, d_dwork);
                if (ds != B.stream)
                {
                    pg_event_record(B.ev_join, ds);
                    pg_stream_wait(B.stream, B.ev_join);
                }
            }
#endif
            HIP_CHECK(hipGetLastError());
        }
#if defined(PG_DENSE_PANELS) || defined(PG_COMPLEX_PANELS)
        if (ndense)
        {
            if (B.opt_count_flops)
                PG_LAUNCH(trsm_flop_count_kernel, dim3((unsigned)ndense), dim3(256), 0, B.stream, d_ftasks, nb, B.d_flops + 2,
                                   B.d_flops + 3);
            B.stats.trsm_dense_tasks += ndense;
        }
#endif
        release_pending_segments();
#if defined(PG_DENSE_PANELS) || defined(PG_COMPLEX_PANELS)
        // the solutions live in the mirrors: bring the sparse records (the authoritative form of a finished block) up
        // to date at once; the mirrors stay valid as MFMA operands
        for (slot_t *s : solved_dense)
        {
            BlockState &st = block_state(s, nb);
            st.mirror_current = true;
            MP.to_sparsify.push_back(mirror_job(s, st.mirror, nb));
            st.sparse_current = true;
        }
        if (!MP.to_sparsify.empty())
            flush_mirror_jobs(nb, MP.to_sparsify, false, true);
#endif
        // one launch serves both kinds; book it under the kind with more tasks, count tasks/bytes exactly
        B.stats.launches[nt >= ng ? 2 : 3]++;
        B.stats.tasks[2] += nt;
        B.stats.tasks[3] += ng;
        B.stats.alg_bytes[2] += by_t;
        B.stats.alg_bytes[3] += by_g;
        if (B.opt_host_mirror)
            for (size_t k = 0; k < take; k++)
                mirror_to_host(list[i + k]->opdst, nb);
        i += take;
    }
    PEND.hold = false;
    flush_pending_getrf(); // (nothing stays held past the call that could have used it)
}

// ---- GETRF -------------------------------------------------------------------------------------------------------
// `gs`: stream the factorisation kernels go to (the main stream, or a side stream that has already been made to wait
// for everything these blocks depend on; the caller joins it back)
// the tiled GETRF kernel (pg_hip_getrf_tiled.h) is the default; PANGULU_HIP_GETRF_TILED=0 selects round 1's kernels
inline bool getrf_tiled_selected()
{
    static const bool on = !(getenv("PANGULU_HIP_GETRF_TILED") && atoi(getenv("PANGULU_HIP_GETRF_TILED")) == 0);
    return on;
}

void launch_getrf(int nb, task_t **list, size_t n, hipStream_t gs, bool defer_join)
{
    HostTimer ht(2);
    const int max_slots = 256;
    if (!B.getrf_scratch || B.nb_cfg != nb)
    {
        B.generation++; // (recorded launches point into the scratch)
        if (B.getrf_scratch)
        {
            HIP_CHECK(hipDeviceSynchronize()); // (factorisations run on side streams too)
            HIP_CHECK(hipFree(B.getrf_scratch));
        }
        HIP_CHECK(hipMalloc((void **)&B.getrf_scratch, std::max(sizeof(val_t), sizeof(double)) * (size_t)nb * nb * max_slots)); // (a slot holds a double image)
        B.getrf_scratch_slots = max_slots;
        B.nb_cfg = nb;
    }
    bool blocked_kernel = false;
#if defined(PG_DENSE_PANELS)
    blocked_kernel = !B.opt_getrf_strict && (nb % 16 == 0) && nb <= GETRF_BLOCKED_ROWS;
#endif
    size_t i = 0;
    while (i < n)
    {
        Segment seg = acquire_segment();
        size_t take = std::min(n - i, (size_t)B.getrf_scratch_slots);
        GetrfTaskD *d_tasks;
        GetrfTaskD *tasks = seg.alloc<GetrfTaskD>(take, &d_tasks);
#if defined(PG_DENSE_PANELS)
        std::vector<double *> lu_images; // dense images that will hold L\\U after this launch
        std::vector<MirrorJobD> deferred; // their sparse records are written by sparsify jobs on the records stream
        bool held = false;                // the launch waits for the next platform call (PendingGetrf)
#endif
#if defined(PG_COMPLEX_PANELS)
        // complex types: diagonal blocks that have a mirror are factorised THERE (zgetrf_planes_kernel), the others by the
        // pattern-driven kernel; PANGULU_HIP_COMPLEX_PANELS=0: all of them by the pattern-driven kernel
        static const bool zpanels_on = !(getenv("PANGULU_HIP_COMPLEX_PANELS") && atoi(getenv("PANGULU_HIP_COMPLEX_PANELS")) == 0);
        std::vector<ZGetrfTaskD> ztasks;
        std::vector<GetrfTaskD> zcount; // their pattern views, for the structural flop count
        std::vector<MirrorJobD> deferred; // sparse records of the blocks factorised in their mirrors: sparsify jobs behind the kernel
#endif
        size_t nsp = 0; // tasks of the pattern-driven / blocked launch
        double by = 0;
        for (size_t k = 0; k < take; k++)
        {
            slot_t *up, *lo;
            diag_halves(list[i + k]->opdst, &up, &lo);
            GetrfTaskD T;
            T.lcp = lo->d_columnpointer;
            T.lri = lo->d_rowindex;
            T.lval = lo->d_value;
            T.urp = up->d_rowpointer;
            T.uci = up->d_columnindex;
            T.uval = up->d_value;
            T.dense = reinterpret_cast<val_t *>(reinterpret_cast<char *>(B.getrf_scratch) + std::max(sizeof(val_t), sizeof(double)) * (size_t)k * nb * nb);
            T.preloaded = 0;
            T.defer_gather = 0;
            T.invert_tiles = 0;
#if defined(PG_COMPLEX_PANELS)
            {
                BlockState &st = block_state(lo, nb);
                double *m = (zpanels_on && !B.opt_getrf_strict && !B.opt_host_mirror && (nb == 128 || nb == 256) && dense_mode_available(nb)) ? obtain_mirror(st, nb) : nullptr;
                if (m)
                {
                    // in the mirror: bring it up to date if the record is ahead, factorise it there, and let a sparsify job write
                    // the record behind the kernel; the image serves the dense solves of this level (ztrsm_planes_kernel)
                    if (!st.mirror_current)
                    {
                        MP.to_densify.push_back(mirror_job(lo, m, nb));
                        st.mirror_current = true;
                    }
                    ztasks.push_back(ZGetrfTaskD{m});
                    zcount.push_back(T);
                    deferred.push_back(mirror_job(lo, m, nb));
                    st.sparse_current = true; // (once the deferred job has run: everything that reads the record waits for it)
                    st.lu_image = true;
                    st.lu_map = false;
                    st.image_halves = 3;
                    by += (2 * SV + 2) * ((double)host_nnz(lo, nb) + host_nnz(up, nb)) + 8.0 * (nb + 1);
                    continue;
                }
                // (no mirror to be had) updates may have accumulated in the block's mirror: the record catches up first,
                // and the mirror is stale once the block is factorised
                if (!st.sparse_current && st.mirror)
                    MP.to_sparsify.push_back(mirror_job(lo, st.mirror, nb));
                st.sparse_current = true;
                st.mirror_current = false;
                st.lu_image = false;
            }
#endif
#if defined(PG_DENSE_PANELS)
            {
                // work on the block's own mirror whenever the pool has one: it may already hold the block (updates
                // accumulated there), and the dense LU it is left with serves the dense TSTRF/GESSM of this level
                BlockState &st = block_state(lo, nb);
                if (blocked_kernel)
                {
                    double *m = dense_mode_available(nb) ? obtain_mirror(st, nb) : nullptr;
                    if (m)
                    {
                        T.dense = reinterpret_cast<val_t *>(m);
                        T.preloaded = (st.mirror_current && !st.sparse_current) ? 1u : 0u;
                        T.invert_tiles = 1;
                        if (B.opt_records_stream && nb <= 256)
                        {
                            T.defer_gather = 1;
                            MirrorJobD J = mirror_job(lo, m, nb);
                            J.diag_tiles = m + (size_t)nb * nb + MIRROR_MAP_BYTES / sizeof(double);
                            deferred.push_back(J);
                        }
                        lu_images.push_back(m);
                        st.lu_image = true;
                        st.lu_map = getrf_tiled_selected();
                        st.image_halves = 3;
                    }
                    else if (!st.sparse_current && st.mirror)
                    {
                        MP.to_sparsify.push_back(mirror_job(lo, st.mirror, nb));
                    }
                }
                else if (!st.sparse_current && st.mirror)
                {
                    MP.to_sparsify.push_back(mirror_job(lo, st.mirror, nb));
                }
                st.sparse_current = true;
                st.mirror_current = false;
            }
#endif
            tasks[nsp++] = T;
            by += (2 * SV + 2) * ((double)host_nnz(lo, nb) + host_nnz(up, nb)) + 8.0 * (nb + 1);
        }
        hipStream_t ks = gs;
#if defined(PG_DENSE_UPDATES)
        if (!MP.to_sparsify.empty())
        {
            flush_mirror_jobs(nb, MP.to_sparsify, false); // (main stream) these blocks must see it: stay on the main stream
            ks = B.stream;
        }
#endif
#if defined(PG_COMPLEX_PANELS)
        ZGetrfTaskD *d_ztasks = nullptr;
        GetrfTaskD *d_zcount = nullptr;
        if (!ztasks.empty())
        {
            if (!MP.to_densify.empty() || !B.opt_records_stream)
                ks = B.stream; // (mirror jobs run on the main stream: the factorisation follows them there)
            if (!MP.to_densify.empty())
                flush_mirror_jobs(nb, MP.to_densify, true);
            ZGetrfTaskD *hz = seg.alloc<ZGetrfTaskD>(ztasks.size(), &d_ztasks);
            GetrfTaskD *hc = seg.alloc<GetrfTaskD>(zcount.size(), &d_zcount);
            if (hc)
                memcpy(hc, zcount.data(), sizeof(GetrfTaskD) * zcount.size());
            if (!hz || !hc)
            {
                fprintf(stderr, "[PanguLU-AMD ERROR] descriptor staging segment too small\n");
                exit(EXIT_FAILURE);
            }
            memcpy(hz, ztasks.data(), sizeof(ZGetrfTaskD) * ztasks.size());
        }
#endif
        commit_segment(seg);
        // (no join with the records stream: its jobs in flight write the records of blocks that are finished, these
        // kernels touch the records of the blocks they factorise)
        {
            LaunchTimer lt(1, ks);
            bool blocked = blocked_kernel;
#if defined(PG_DENSE_PANELS)
            if (blocked)
            {
                size_t lds = sizeof(double) * (2 * GETRF_PANEL * (size_t)(nb + 2) + GETRF_PANEL * GETRF_PANEL) + sizeof(u32) * 2 * (size_t)(nb + 1);
                static size_t lds_allowed = 0;
                if (lds > lds_allowed)
                {
                    HIP_CHECK(hipFuncSetAttribute((const void *)getrf_blocked_f64_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    HIP_CHECK(hipFuncSetAttribute((const void *)getrf_blocked_f64_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    lds_allowed = lds;
                }
                static const bool debug_stamps = getenv("PANGULU_HIP_DEBUG_GETRF") != nullptr;
                // (measured: 64.2 ms per factorisation of the bench matrix with the 512-thread variant from 129 blocks against 64.6 ms
                // without -- both kernels slow down when they share CUs; off by default)
                static const long narrow_from = getenv("PANGULU_HIP_GETRF_NARROW_FROM") ? atol(getenv("PANGULU_HIP_GETRF_NARROW_FROM")) : 1 << 30;
                static const bool lookahead_kernel = !(getenv("PANGULU_HIP_GETRF_LOOKAHEAD") && atoi(getenv("PANGULU_HIP_GETRF_LOOKAHEAD")) == 0);
                static const bool tiled_kernel = getrf_tiled_selected();
                if (tiled_kernel)
                {
                    // static tile ownership + a dedicated factorisation wavefront (pg_hip_getrf_tiled.h)
                    const size_t lds_t = gt_lds_bytes(nb);
                    static size_t t_allowed = 0;
                    if (lds_t > t_allowed)
                    {
                        HIP_CHECK(hipFuncSetAttribute((const void *)getrf_tiled_f64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
                        t_allowed = lds_t;
                    }
                    // (held for the chase when it could serve the dense solves of its level: see PendingGetrf)
                    static const bool chase_on = getenv("PANGULU_HIP_CHASE") && atoi(getenv("PANGULU_HIP_CHASE")) != 0; // (off by default: see getrf_trsm_chase_kernel)
                    bool all_images = !lu_images.empty() && lu_images.size() == take;
                    for (size_t k = 0; k < take && all_images; k++)
                        all_images = tasks[k].invert_tiles && tasks[k].defer_gather;
                    // (near the root only: a level with many diagonal blocks is bound by throughput, and there the two-in-one launch costs
                    //  more than the chain it removes -- PANGULU_HIP_CHASE_MAX_GETRF)
                    static const size_t chase_max = getenv("PANGULU_HIP_CHASE_MAX_GETRF") ? (size_t)atol(getenv("PANGULU_HIP_CHASE_MAX_GETRF")) : 4;
                    if (chase_on && REC.mode != 0 && ks == B.stream && i == 0 && take == n && take <= chase_max && all_images && !debug_stamps && !B.opt_profile &&
                        !B.opt_host_mirror && (nb == 128 || nb == 256))
                    {
                        held = true;
                        PEND.nb = nb;
                        PEND.take = take;
                        PEND.d_tasks = d_tasks;
                        PEND.images.assign(lu_images.begin(), lu_images.end());
                        if (!B.d_progress)
                        {
                            B.generation++;
                            HIP_CHECK(hipMalloc((void **)&B.d_progress, sizeof(unsigned) * PROGRESS_WORDS));
                            HIP_CHECK(hipMemset(B.d_progress, 0, sizeof(unsigned) * PROGRESS_WORDS));
                        }
                        if (B.progress_next + take > PROGRESS_WORDS)
                            B.progress_next = 0;
                        PEND.d_progress = B.d_progress + B.progress_next;
                        B.progress_next += take;
                        unsigned long long *fc = B.d_flops + 1;
                        const unsigned ntake = (unsigned)take;
                        PEND.plain = [=]()
                        { PG_LAUNCH(getrf_tiled_f64_kernel, dim3(ntake), dim3(GT_THREADS), lds_t, ks, d_tasks, nb, fc, (unsigned long long *)nullptr); };
                    }
                    else
                        PG_LAUNCH(getrf_tiled_f64_kernel, dim3((unsigned)take), dim3(GT_THREADS), lds_t, ks, d_tasks, nb, B.d_flops + 1,
                                           debug_stamps ? B.d_flops + 8 : nullptr);
                }
                else if (lookahead_kernel)
                {
                    const size_t lds_la = sizeof(double) * (4 * GETRF_PANEL * (size_t)(nb + 2) + GETRF_PANEL * GETRF_PANEL) + sizeof(u32) * (2 * (size_t)(nb + 1) + 4);
                    static size_t la_allowed = 0;
                    if (lds_la > la_allowed)
                    {
                        HIP_CHECK(hipFuncSetAttribute((const void *)getrf_lookahead_f64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_la));
                        la_allowed = lds_la;
                    }
                    PG_LAUNCH(getrf_lookahead_f64_kernel, dim3((unsigned)take), dim3(1024), lds_la, ks, d_tasks, nb, B.d_flops + 1,
                                       debug_stamps ? B.d_flops + 8 : nullptr);
                }
                else if ((long)take >= narrow_from)
                    PG_LAUNCH(getrf_blocked_f64_kernel<512>, dim3((unsigned)take), dim3(512), lds, ks, d_tasks, nb,
                                       B.d_flops + 1, debug_stamps ? B.d_flops + 8 : nullptr);
                else
                    PG_LAUNCH(getrf_blocked_f64_kernel<1024>, dim3((unsigned)take), dim3(1024), lds, ks, d_tasks, nb,
                                       B.d_flops + 1, debug_stamps ? B.d_flops + 8 : nullptr);
            }
#endif
            if (!blocked && nsp)
            {
                size_t lds = (sizeof(val_t) * 2 + sizeof(u16) * 2) * (size_t)nb;
                PG_LAUNCH(getrf_kernel, dim3((unsigned)nsp), dim3(GETRF_THREADS), lds, ks, d_tasks, nb, B.d_flops + 1);
            }
#if defined(PG_COMPLEX_PANELS)
            if (!ztasks.empty())
            {
                const size_t lds_z = sizeof(double) * 4 * ZP_PANEL * (size_t)nb;
                static size_t z_allowed = 0;
                if (lds_z > z_allowed)
                {
                    HIP_CHECK(hipFuncSetAttribute((const void *)zgetrf_planes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_z));
                    z_allowed = lds_z;
                }
                PG_LAUNCH(zgetrf_planes_kernel, dim3((unsigned)ztasks.size()), dim3(ZG_THREADS), lds_z, ks, (const ZGetrfTaskD *)d_ztasks, nb);
                if (B.opt_count_flops)
                    PG_LAUNCH(getrf_flop_count_kernel, dim3((unsigned)ztasks.size()), dim3(256), 0, ks, (const GetrfTaskD *)d_zcount, nb, B.d_flops + 1);
                B.zgetrf_tasks += ztasks.size();
            }
#endif
            HIP_CHECK(hipGetLastError());
        }
#if defined(PG_COMPLEX_PANELS)
        if (!deferred.empty())
            pg_event_record(B.ev_rec_fork, ks); // behind the factorisation
#endif
#if defined(PG_DENSE_PANELS)
        if (held)
        {
            // (everything that follows the launch follows it when it is made: PendingGetrf)
            PEND.post = [=]() mutable
            {
                if (!deferred.empty())
                    pg_event_record(B.ev_rec_fork, ks); // behind the factorisation
                release_pending_segments(ks);
                if (!deferred.empty())
                    flush_mirror_jobs(nb, deferred, false, true, nullptr, true);
                B.stats.launches[1]++;
                B.stats.tasks[1] += take;
                B.stats.alg_bytes[1] += by;
            };
            PEND.active = true;
            return; // (take == n)
        }
        if (!deferred.empty())
            pg_event_record(B.ev_rec_fork, ks); // behind the factorisation
#endif
        if (ks != B.stream)
        {
            pg_event_record(B.ev_join3, ks);
            if (defer_join && !B.opt_host_mirror)
                B.getrf_join_pending = true; // the caller makes the main stream wait once its own kernels are queued
            else
                pg_stream_wait(B.stream, B.ev_join3);
        }
        release_pending_segments(ks); // (the descriptors are read on ks, which the main stream may not have joined yet)
#if defined(PG_DENSE_PANELS) || defined(PG_COMPLEX_PANELS)
        if (!deferred.empty())
            flush_mirror_jobs(nb, deferred, false, true, nullptr, true);
#endif
        B.stats.launches[1]++;
        B.stats.tasks[1] += take;
        B.stats.alg_bytes[1] += by;
        if (B.opt_host_mirror)
            for (size_t k = 0; k < take; k++)
            {
                slot_t *up, *lo;
                diag_halves(list[i + k]->opdst, &up, &lo);
                mirror_to_host(up, nb);
                mirror_to_host(lo, nb);
            }
        i += take;
    }
}

void check_lds_budget(int nb)
{
    size_t need = sizeof(val_t) * (size_t)nb * SSSSM_WAVES;
    if (need > 160 * 1024 || nb > 8192)
    {
        fprintf(stderr, "[PanguLU-AMD ERROR] nb = %d needs %zu bytes of LDS per workgroup; reduce init_options.nb\n", nb, need);
        exit(EXIT_FAILURE);
    }
}

// one dependency-free run of tasks: one launch per kernel class
void process_run(int nb, task_t *tasks, size_t n, std::vector<task_t *> &getrf, std::vector<task_t *> &trsm, std::vector<task_t *> &ssssm)
{
    getrf.clear();
    trsm.clear();
    ssssm.clear();
    for (size_t i = 0; i < n; i++)
    {
        switch (tasks[i].kernel_id)
        {
        case PANGULU_TASK_GETRF:
            getrf.push_back(&tasks[i]);
            break;
        case PANGULU_TASK_TSTRF:
        case PANGULU_TASK_GESSM:
            trsm.push_back(&tasks[i]);
            break;
        case PANGULU_TASK_SSSSM:
            ssssm.push_back(&tasks[i]);
            break;
        default:
            fprintf(stderr, "[PanguLU-AMD ERROR] unknown kernel id %d\n", (int)tasks[i].kernel_id);
            exit(EXIT_FAILURE);
        }
    }
    if (!(getrf.empty() && ssssm.empty() && !trsm.empty()))
        flush_pending_getrf(); // (only the solves of its own level may chase a held factorisation)
    if (!B.opt_assume_independent && ssssm.size() > 1)
    {
        // updates of one destination must be adjacent (they share one LDS accumulator pass)
        std::stable_sort(ssssm.begin(), ssssm.end(), [](const task_t *x, const task_t *y)
                         {
                             return canon_dst(x->opdst) < canon_dst(y->opdst); });
    }
    // The factorisations, the solves and the updates of one run are independent of each other: the GETRFs (a handful of
    // workgroups, latency-bound) go to a side stream that waits only for what was queued before this run and run beside
    // the update and TSTRF/GESSM kernels; the main stream joins at the end of the run.
    // (with CU-masked bulk streams a run of GETRFs alone goes there as well: a leaf level has one workgroup per CU)
    // blocks that update launches on the background stream may still be writing: wait before anything of this run touches them
    if (B.bg_active)
    {
        bool hit = false;
        for (size_t i = 0; i < n && !hit; i++)
            for (const slot_t *sl : {(const slot_t *)tasks[i].opdst, (const slot_t *)tasks[i].op1, (const slot_t *)tasks[i].op2})
                if (sl && B.bg_tiles.count(block_key_any(sl)))
                {
                    hit = true;
                    break;
                }
        if (hit)
            join_background(B.stream);
    }
    // Look-ahead call of the scheduler (diagonal factorisations + every update queued anywhere, independent of each other):
    // the GETRFs are the critical path and stay on the main stream, the updates -- trailing-matrix work of the previous
    // level -- go to the background stream and are NOT joined at the end: the panel solves of the next call run beside them.
    const bool background = B.opt_background_updates && B.opt_two_streams && B.opt_assume_independent && !getrf.empty() && !ssssm.empty() &&
                            trsm.empty() && !B.bulk_streams_masked && !B.opt_profile;
    if (background)
    {
        launch_ssssm(nb, ssssm.data(), ssssm.size(), true);
        launch_getrf(nb, getrf.data(), getrf.size(), B.stream, false);
        return;
    }
    bool side = B.opt_two_streams && !getrf.empty() && (!trsm.empty() || !ssssm.empty() || B.bulk_streams_masked);
    if (side)
    {
        pg_event_record(B.ev_fork3, B.stream);
        pg_stream_wait(B.stream3, B.ev_fork3);
        B.getrf_join_pending = false;
        launch_getrf(nb, getrf.data(), getrf.size(), B.stream3, true);
    }
    launch_ssssm(nb, ssssm.data(), ssssm.size());
    launch_trsm(nb, trsm.data(), trsm.size());
    if (side)
    {
        if (B.getrf_join_pending)
            pg_stream_wait(B.stream, B.ev_join3);
        B.getrf_join_pending = false;
    }
    else
        launch_getrf(nb, getrf.data(), getrf.size(), B.stream, false);
}

} // namespace

// =================================================================================================================
// C-ABI
// =================================================================================================================
extern "C"
{

    void pangulu_platform_0201001_malloc(void **platform_address, size_t size)
    {
        ensure_ready();
        HIP_CHECK(hipMalloc(platform_address, size ? size : 16));
    }

    void pangulu_platform_0201001_malloc_pinned(void **platform_address, size_t size)
    {
        ensure_ready();
        HIP_CHECK(hipHostMalloc(platform_address, size ? size : 16, hipHostMallocDefault));
    }

    void pangulu_platform_0201001_synchronize(void)
    {
        ensure_ready();
        flush_pending_getrf_locked();
        join_records(B.stream);
        join_background(B.stream);
        HIP_CHECK(hipStreamSynchronize(B.stream));
    }

    void pangulu_platform_0201001_memset(void *s, int c, size_t n)
    {
        ensure_ready();
        flush_pending_getrf_locked();
        HIP_CHECK(hipMemsetAsync(s, c, n, B.stream));
        HIP_CHECK(hipStreamSynchronize(B.stream));
    }

    void pangulu_platform_0201001_create_stream(void **stream)
    {
        ensure_ready();
        HIP_CHECK(hipStreamCreateWithFlags((hipStream_t *)stream, hipStreamNonBlocking));
    }

    static hipMemcpyKind kind_of(unsigned int kind)
    {
        switch (kind)
        {
        case 0:
            return hipMemcpyHostToDevice;
        case 1:
            return hipMemcpyDeviceToHost;
        case 2:
            return hipMemcpyDeviceToDevice;
        default:
            fprintf(stderr, "[PanguLU-AMD ERROR] invalid memcpy kind %u\n", kind);
            exit(EXIT_FAILURE);
        }
    }

    void pangulu_platform_0201001_memcpy(void *dst, const void *src, size_t count, unsigned int kind)
    {
        ensure_ready();
        flush_pending_getrf_locked();
        // ordered after everything queued on the back-end stream, complete on return
        join_records(B.stream);
        join_background(B.stream);
        HIP_CHECK(hipMemcpyAsync(dst, src, count, kind_of(kind), B.stream));
        HIP_CHECK(hipStreamSynchronize(B.stream));
    }

    void pangulu_platform_0201001_memcpy_async(void *dst, const void *src, size_t count, unsigned int kind, void *stream)
    {
        ensure_ready();
        flush_pending_getrf_locked();
        // stream == NULL is what the reference host passes from its receive thread
        // (src/pangulu_communication.c:1850,1880): use the back-end stream so later kernels are ordered behind it
        hipStream_t s = stream ? (hipStream_t)stream : B.stream;
        if (kind != 0)
        {
            join_records(s); // (uploads of received blocks write receive slots, which no sparsify job touches)
            join_background(s);
        }
        HIP_CHECK(hipMemcpyAsync(dst, src, count, kind_of(kind), s));
        if (!stream)
            HIP_CHECK(hipStreamSynchronize(s)); // the source is pageable host memory the caller may reuse at once
    }

    void pangulu_platform_0201001_free(void *devptr)
    {
        flush_pending_getrf_locked();
        if (!devptr)
            return;
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, devptr) == hipSuccess && attr.type == hipMemoryTypeHost)
        {
            HIP_CHECK(hipHostFree(devptr));
            return;
        }
        HIP_CHECK(hipFree(devptr));
    }

    void pangulu_platform_0201001_get_device_num(int *device_num)
    {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess)
            n = 0;
        *device_num = n;
    }

    // CPUs of the NUMA node the device hangs off (sysfs), intersected with what the thread may run on.
    static bool cpus_near_device(int device, cpu_set_t *out)
    {
        char bdf[64] = {0};
        if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess)
        {
            (void)hipGetLastError();
            return false;
        }
        for (char *c = bdf; *c; c++)
            *c = (char)tolower(*c);
        char path[160];
        snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
        FILE *f = fopen(path, "r");
        int node = -1;
        if (!f || fscanf(f, "%d", &node) != 1)
            node = -1;
        if (f)
            fclose(f);
        if (node < 0)
            return false;
        snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
        f = fopen(path, "r");
        if (!f)
            return false;
        char list[1024] = {0};
        const bool got = fgets(list, sizeof(list), f) != nullptr;
        fclose(f);
        if (!got)
            return false;
        cpu_set_t allowed, want;
        CPU_ZERO(&want);
        if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0)
            return false;
        for (char *tok = strtok(list, ",\n"); tok; tok = strtok(nullptr, ",\n"))
        {
            int lo = 0, hi = 0;
            const int k = sscanf(tok, "%d-%d", &lo, &hi);
            if (k < 1)
                continue;
            if (k == 1)
                hi = lo;
            for (int c = lo; c <= hi && c < CPU_SETSIZE; c++)
                if (CPU_ISSET(c, &allowed))
                    CPU_SET(c, &want);
        }
        if (CPU_COUNT(&want) == 0)
            return false;
        *out = want;
        return true;
    }

    // Keep the calling thread (and the threads it creates) on the CPUs next to the device while `enable`, give it back its
    // old mask otherwise.  The host side of the factorisation is two latency-critical threads (scheduler and launcher) that
    // write task descriptors into pinned host memory the kernels read in place: on the two-socket bench host a
    // factorisation takes 46.0-48.1 ms with them on the device's NUMA node, 50-60 ms (and single steps up to 74 ms) wherever
    // the OS puts them, 62 ms on the other socket.  The reference pins its threads too (pangulu_bind_to_core,
    // src/pangulu_thread.c:3-12).  PANGULU_AMD_BIND_NUMA=0 turns it off.  Returns 0 when the mask was changed / restored.
    int pangulu_platform_0201001_bind_near_device(int enable)
    {
        static thread_local cpu_set_t saved;
        static thread_local int depth = 0; // bind / unbind pairs nest: only the outermost pair changes the mask
        static const bool off = getenv("PANGULU_AMD_BIND_NUMA") && atoi(getenv("PANGULU_AMD_BIND_NUMA")) == 0;
        if (off)
            return 1;
        if (!enable)
        {
            if (depth == 0)
                return 1;
            if (--depth > 0)
                return 0; // (an outer pair is still active: stay where we are)
            return sched_setaffinity(0, sizeof(saved), &saved) == 0 ? 0 : 1;
        }
        if (depth > 0)
        {
            depth++;
            return 0; // (nested: already there)
        }
        cpu_set_t want;
        if (!cpus_near_device(B.device, &want))
            return 1;
        if (sched_getaffinity(0, sizeof(saved), &saved) != 0)
            return 1;
        if (sched_setaffinity(0, sizeof(want), &want) != 0)
            return 1;
        depth = 1;
        return 0;
    }

    void pangulu_platform_0201001_set_default_device(int device_num)
    {
        B.device = device_num;
        HIP_CHECK(hipSetDevice(device_num));
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device_num));
        pangulu_gpu_shared_mem_size = (int)prop.sharedMemPerBlock;
        // (the pinned descriptor segments are allocated here: from the device's NUMA node)
        const int bound = pangulu_platform_0201001_bind_near_device(1);
        ensure_ready();
        if (bound == 0)
            pangulu_platform_0201001_bind_near_device(0);
    }

    void pangulu_platform_0201001_get_device_name(char *name, int device_num)
    {
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device_num));
        strcpy(name, prop.name);
    }

    void pangulu_platform_0201001_get_device_memory_usage(size_t *used_byte)
    {
        size_t free_b = 0, total_b = 0;
        HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        *used_byte = total_b - free_b;
    }

    void pangulu_platform_0201001_hybrid_batched(pangulu_inblock_idx nb, pangulu_uint64_t ntask, pangulu_task_t *tasks)
    {
        ensure_ready();
        if (ntask == 0)
            return;
        std::lock_guard<std::mutex> g(B.mutex);
        HostTimer ht_call(5);
        HIP_CHECK(hipSetDevice(B.device));
        check_lds_budget(nb);
        static thread_local std::vector<task_t *> l_getrf, l_trsm, l_ssssm;
        if (B.opt_assume_independent)
        {
            process_run(nb, tasks, (size_t)ntask, l_getrf, l_trsm, l_ssssm);
            release_pending_segments();
            return;
        }
        // The reference executes the array serially (...0201000.cu:875-898).  Keep that meaning: cut the array
        // wherever a task touches a block an earlier task of the current run writes, and batch inside each run.
        std::unordered_map<const slot_t *, int> written; // block -> kernel class of its writer in this run
        auto canon = [](const slot_t *s) -> const slot_t *
        {
            if (s && s->brow_pos == s->bcol_pos && s->is_upper && s->related_block)
                return s->related_block;
            return s;
        };
        size_t run_begin = 0;
        for (size_t i = 0; i < (size_t)ntask; i++)
        {
            const task_t &t = tasks[i];
            const slot_t *d = canon(t.opdst), *a = canon(t.op1), *b = canon(t.op2);
            bool hazard = false;
            auto wd = written.find(d);
            if (wd != written.end() && !(t.kernel_id == PANGULU_TASK_SSSSM && wd->second == PANGULU_TASK_SSSSM))
                hazard = true;
            if (a && written.count(a))
                hazard = true;
            if (b && written.count(b))
                hazard = true;
            if (hazard)
            {
                process_run(nb, tasks + run_begin, i - run_begin, l_getrf, l_trsm, l_ssssm);
                run_begin = i;
                written.clear();
            }
            written[d] = t.kernel_id;
        }
        process_run(nb, tasks + run_begin, (size_t)ntask - run_begin, l_getrf, l_trsm, l_ssssm);
        release_pending_segments();
    }

    void pangulu_platform_0201001_ssssm_batched(pangulu_inblock_idx nb, pangulu_uint64_t ntask, pangulu_task_t *tasks)
    {
        flush_pending_getrf_locked();
        // as the reference's dispatcher does (src/pangulu_kernel_interface.c:302), kernel ids in the array decide
        pangulu_platform_0201001_hybrid_batched(nb, ntask, tasks);
    }

    static void single_task(pangulu_inblock_idx nb, int kernel, slot_t *dst, slot_t *op1, slot_t *op2)
    {
        task_t t;
        memset(&t, 0, sizeof(t));
        t.kernel_id = (pangulu_int16_t)kernel;
        t.row = dst->brow_pos;
        t.col = dst->bcol_pos;
        t.opdst = dst;
        t.op1 = op1;
        t.op2 = op2;
        pangulu_platform_0201001_hybrid_batched(nb, 1, &t);
    }

    void pangulu_platform_0201001_getrf(pangulu_inblock_idx nb, pangulu_storage_slot_t *opdst, int tid)
    {
        (void)tid;
        single_task(nb, PANGULU_TASK_GETRF, opdst, nullptr, nullptr);
    }
    void pangulu_platform_0201001_tstrf(pangulu_inblock_idx nb, pangulu_storage_slot_t *opdst, pangulu_storage_slot_t *opdiag, int tid)
    {
        (void)tid;
        single_task(nb, PANGULU_TASK_TSTRF, opdst, opdiag, nullptr);
    }
    void pangulu_platform_0201001_gessm(pangulu_inblock_idx nb, pangulu_storage_slot_t *opdst, pangulu_storage_slot_t *opdiag, int tid)
    {
        (void)tid;
        single_task(nb, PANGULU_TASK_GESSM, opdst, opdiag, nullptr);
    }
    void pangulu_platform_0201001_ssssm(pangulu_inblock_idx nb, pangulu_storage_slot_t *opdst, pangulu_storage_slot_t *op1,
                                        pangulu_storage_slot_t *op2, int tid)
    {
        (void)tid;
        single_task(nb, PANGULU_TASK_SSSSM, opdst, op1, op2);
    }

    void pangulu_platform_0201001_spmv(pangulu_inblock_idx nb, pangulu_storage_slot_t *a, calculate_type *x, calculate_type *y)
    {
        ensure_ready();
        flush_pending_getrf_locked();
        join_records(B.stream);
        hipLaunchKernelGGL(spmv_kernel, dim3(1), dim3(256), 0, B.stream, (int)nb, a->d_columnpointer, a->d_rowindex, a->d_value, x, y);
        HIP_CHECK(hipGetLastError());
    }

    void pangulu_platform_0201001_vecadd(pangulu_int64_t length, calculate_type *bval, calculate_type *xval)
    {
        ensure_ready();
        flush_pending_getrf_locked();
        if (length <= 0)
            return;
        hipLaunchKernelGGL(vecadd_kernel, dim3((unsigned)((length + 255) / 256)), dim3(256), 0, B.stream, (long long)length, bval, xval);
        HIP_CHECK(hipGetLastError());
    }

    void pangulu_platform_0201001_sptrsv(pangulu_inblock_idx nb, pangulu_storage_slot_t *s, calculate_type *xval, pangulu_int64_t uplo)
    {
        ensure_ready();
        flush_pending_getrf_locked();
        size_t lds = sizeof(val_t) * (size_t)nb;
        join_records(B.stream);
        if (uplo == PANGULU_LOWER)
            hipLaunchKernelGGL(sptrsv_kernel, dim3(1), dim3(256), lds, B.stream, (int)nb, s->d_columnpointer, s->d_rowindex, s->d_value, xval, 0);
        else
            hipLaunchKernelGGL(sptrsv_kernel, dim3(1), dim3(256), lds, B.stream, (int)nb, s->d_rowpointer, s->d_columnindex, s->d_value, xval, 1);
        HIP_CHECK(hipGetLastError());
    }

    // ---- markers: "everything queued on the back-end up to now" as a waitable handle -----------------------------
    // (a ring of events, never freed: a handle that is re-recorded meanwhile simply completes later)
    void *pangulu_platform_0201001_marker_record(void)
    {
        ensure_ready();
        std::lock_guard<std::mutex> g(B.mutex);
        flush_pending_getrf();
        HIP_CHECK(hipSetDevice(B.device));
        static std::vector<hipEvent_t> ring;
        static size_t next = 0;
        if (ring.empty())
        {
            ring.resize(1024);
            for (hipEvent_t &e : ring)
                HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        hipEvent_t e = ring[next];
        next = (next + 1) % ring.size();
        join_records(B.stream); // a marker stands for "the blocks finished so far can be sent": their records included
        join_background(B.stream); // ... and for "the slots whose last consumer has been queued may be reused": background updates too
        HIP_CHECK(hipEventRecord(e, B.stream)); // (side streams have been joined into the main stream by every call)
        return (void *)e;
    }

    int pangulu_platform_0201001_marker_done(void *marker)
    {
        hipError_t r = hipEventQuery((hipEvent_t)marker);
        if (r == hipSuccess)
            return 1;
        if (r != hipErrorNotReady)
            HIP_CHECK(r);
        (void)hipGetLastError();
        return 0;
    }

    void pangulu_platform_0201001_marker_wait(void *marker)
    {
        // (called from the transport's sender thread: no back-end state is touched here -- marker_record has launched a held
        //  factorisation before it recorded the event this waits for)
        HIP_CHECK(hipSetDevice(B.device));
        HIP_CHECK(hipEventSynchronize((hipEvent_t)marker));
    }

    void pangulu_platform_0201001_prepare_diag(pangulu_inblock_idx nb, pangulu_storage_slot_t *diag)
    {
        ensure_ready();
        std::lock_guard<std::mutex> g(B.mutex);
        flush_pending_getrf();
        slot_t *up, *lo;
        diag_halves(diag, &up, &lo);
        (void)get_diag_aux(up, nb);
    }

    void pangulu_platform_0201001_prepare_blocks(pangulu_inblock_idx nb, pangulu_uint64_t nslot, pangulu_storage_slot_t **slots)
    {
        flush_pending_getrf_locked();
#if defined(PG_DENSE_UPDATES)
        if (nb > 256 || nb % 16 != 0 || nslot == 0)
            return;
        static const bool enabled = !(getenv("PANGULU_HIP_OCCUPANCY_SUMMARIES") && atoi(getenv("PANGULU_HIP_OCCUPANCY_SUMMARIES")) == 0);
        if (!enabled)
            return;
        ensure_ready();
        std::lock_guard<std::mutex> g(B.mutex);
        struct Occ
        {
            unsigned short a[2], b[2], rows, cols, map[16], map_t[16];
        };
        std::vector<Occ> occ((size_t)nslot);
        // the patterns are walked by a few threads (half a billion entries for the bench matrix), the table is filled by one
        const unsigned nthr = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nthr; t++)
            pool.emplace_back([&, t]()
                              {
                                  for (size_t i = t; i < (size_t)nslot; i += nthr)
                                  {
                                      const slot_t *s = slots[i];
                                      unsigned short m[16] = {0};
                                      const u32 *cp = s->columnpointer;
                                      const u16 *ri = s->rowindex;
                                      for (int c = 0; c < (int)nb; c++)
                                      {
                                          unsigned short bits = 0;
                                          for (u32 p = c == 0 ? 0u : cp[c]; p < cp[c + 1]; p++)
                                              bits |= (unsigned short)(1u << (ri[p] >> 4));
                                          m[c >> 4] |= bits;
                                      }
                                      Occ o;
                                      memset(&o, 0, sizeof(o));
                                      memcpy(o.map, m, sizeof(o.map));
                                      for (int c = 0; c < 16; c++)
                                          for (int r = 0; r < 16; r++)
                                              if ((m[c] >> r) & 1)
                                                  o.map_t[r] |= (unsigned short)(1u << c);
                                      for (int sl = 0; sl < (int)nb / 16; sl++)
                                      {
                                          if (m[sl] & 0x00FF)
                                              o.a[0] |= (unsigned short)(1u << sl);
                                          if (m[sl] & 0xFF00)
                                              o.a[1] |= (unsigned short)(1u << sl);
                                          o.b[sl >> 3] |= m[sl];
                                          o.rows |= m[sl];
                                          if (m[sl])
                                              o.cols |= (unsigned short)(1u << sl);
                                      }
                                      occ[i] = o;
                                  } });
        for (auto &th : pool)
            th.join();
        for (size_t i = 0; i < (size_t)nslot; i++)
        {
            BlockState &st = block_state(slots[i], (int)nb);
            st.occ_valid = true;
            st.occ_a[0] = occ[i].a[0];
            st.occ_a[1] = occ[i].a[1];
            st.occ_b[0] = occ[i].b[0];
            st.occ_b[1] = occ[i].b[1];
            st.occ_rows = occ[i].rows;
            st.occ_cols = occ[i].cols;
            memcpy(st.occ_map, occ[i].map, sizeof(st.occ_map));
            memcpy(st.occ_map_t, occ[i].map_t, sizeof(st.occ_map_t));
        }
#else
        (void)nb;
        (void)nslot;
        (void)slots;
#endif
    }

    int pangulu_platform_0201001_set_option(int option, long long value)
    {
        switch (option)
        {
        case PANGULU_HIP_OPT_HOST_MIRROR:
            B.opt_host_mirror = value;
            return 0;
        case PANGULU_HIP_OPT_DENSE_THRESHOLD_PERMILLE:
            B.opt_dense_permille = value;
            return 0;
        case PANGULU_HIP_OPT_PROFILE:
            B.opt_profile = value;
            return 0;
        case PANGULU_HIP_OPT_ASSUME_INDEPENDENT:
            B.opt_assume_independent = value;
            return 0;
        case PANGULU_HIP_OPT_GETRF_STRICT_ORDER:
            B.opt_getrf_strict = value;
            return 0;
        case PANGULU_HIP_OPT_COUNT_FLOPS:
            B.opt_count_flops = value;
            return 0;
        case PANGULU_HIP_OPT_SSSSM_GROUP_CHUNK:
            B.opt_group_chunk = value;
            return 0;
        case PANGULU_HIP_OPT_TRSM_DENSE_PERMILLE:
            B.opt_trsm_dense_permille = value;
            return 0;
        case PANGULU_HIP_OPT_TWO_STREAMS:
            B.opt_two_streams = value;
            return 0;
        case PANGULU_HIP_OPT_SMALL_LAUNCH_TASKS:
            B.opt_small_launch_tasks = value;
            return 0;
        case PANGULU_HIP_OPT_XCD_SWIZZLE:
        {
            ensure_ready();
            int v = value ? 1 : 0;
            HIP_CHECK(hipStreamSynchronize(B.stream));
            HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_xcd_swizzle), &v, sizeof(v)));
            return 0;
        }
        case PANGULU_HIP_OPT_RESET_BLOCK_STATE:
        {
            std::lock_guard<std::mutex> g(B.mutex);
            reset_block_states();
            return 0;
        }
        case PANGULU_HIP_OPT_BACKGROUND_UPDATES:
            B.opt_background_updates = value;
            return 0;
        case PANGULU_HIP_OPT_FRONT_STAGES:
            B.opt_front_stages = value;
            return 0;
        case PANGULU_HIP_OPT_TILES_STAGES:
            B.opt_tiles_stages = value;
            return 0;
        case PANGULU_HIP_OPT_RECORDS_STREAM:
        {
            ensure_ready();
            std::lock_guard<std::mutex> g(B.mutex);
            join_records(B.stream); // (jobs already on the records stream are joined before the switch takes effect)
            B.opt_records_stream = value;
            return 0;
        }
        default:
            return 1;
        }
    }

    // Level-scheduled block triangular solve (see block_trsv_level_kernel).  `x` is a HOST vector of nbk*nb values: copied
    // to the device, swept forward (L, unit diagonal) or backward (U), copied back.  rows[level_ptr[l] .. level_ptr[l+1]) are
    // the block rows of level l; row r's off-diagonal blocks are blk_slots / blk_bcol[rows[r].first .. + rows[r].nblk).
    void pangulu_platform_0201001_block_trsv(pangulu_inblock_idx nb, int upper, pangulu_uint64_t nlevel, const pangulu_uint64_t *level_ptr,
                                             const pangulu_hip_solve_row_t *rows, pangulu_storage_slot_t *const *blk_slots,
                                             const pangulu_exblock_idx *blk_bcol, calculate_type *x, pangulu_uint64_t xlen)
    {
        ensure_ready();
        std::lock_guard<std::mutex> g(B.mutex);
        flush_pending_getrf();
        HIP_CHECK(hipSetDevice(B.device));
        join_records(B.stream); // the sparse records of finished blocks are written on the records stream
        join_background(B.stream);
        const size_t nrow = (size_t)level_ptr[nlevel];
        size_t nblk = 0;
        for (size_t r = 0; r < nrow; r++)
            nblk += rows[r].nblk;
        std::vector<SolveRowD> hr(std::max<size_t>(nrow, 1));
        std::vector<SolveBlkD> hb(std::max<size_t>(nblk, 1));
        std::vector<size_t> blk_level_ptr((size_t)nlevel + 1, 0);
        size_t o = 0;
        for (size_t l = 0; l < (size_t)nlevel; l++)
        {
            for (size_t r = (size_t)level_ptr[l]; r < (size_t)level_ptr[l + 1]; r++)
            {
                const slot_t *d = rows[r].diag;
                hr[r].brow = rows[r].brow;
                hr[r].nblk = rows[r].nblk;
                hr[r].first = o;
                hr[r].dptr = upper ? d->d_rowpointer : d->d_columnpointer;
                hr[r].didx = upper ? d->d_columnindex : d->d_rowindex;
                hr[r].dval = d->d_value;
                for (size_t b = 0; b < rows[r].nblk; b++, o++)
                {
                    const slot_t *sb = blk_slots[rows[r].first + b];
                    hb[o].cp = sb->d_columnpointer;
                    hb[o].ri = sb->d_rowindex;
                    hb[o].val = sb->d_value;
                    hb[o].bcol = blk_bcol[rows[r].first + b];
                    hb[o].brow = rows[r].brow;
                }
            }
            blk_level_ptr[l + 1] = o;
        }
        SolveRowD *d_rows = nullptr;
        SolveBlkD *d_blks = nullptr;
        val_t *d_x = nullptr;
        HIP_CHECK(hipMalloc((void **)&d_rows, sizeof(SolveRowD) * hr.size()));
        HIP_CHECK(hipMalloc((void **)&d_blks, sizeof(SolveBlkD) * hb.size()));
        HIP_CHECK(hipMalloc((void **)&d_x, sizeof(val_t) * (size_t)xlen));
        HIP_CHECK(hipMemcpyAsync(d_rows, hr.data(), sizeof(SolveRowD) * hr.size(), hipMemcpyHostToDevice, B.stream));
        HIP_CHECK(hipMemcpyAsync(d_blks, hb.data(), sizeof(SolveBlkD) * hb.size(), hipMemcpyHostToDevice, B.stream));
        HIP_CHECK(hipMemcpyAsync(d_x, x, sizeof(val_t) * (size_t)xlen, hipMemcpyHostToDevice, B.stream));
        const size_t lds = sizeof(val_t) * (size_t)nb;
        // round 4 kernels (PANGULU_HIP_SOLVE_CHUNKED=0: the column-by-column ones): chunks of `ch` columns of a diagonal half
        // through at most 96 KB of LDS
        static const bool chunked_on = !(getenv("PANGULU_HIP_SOLVE_CHUNKED") && atoi(getenv("PANGULU_HIP_SOLVE_CHUNKED")) == 0);
        const size_t per_col = 2 * (size_t)nb * (sizeof(val_t) + sizeof(u16)); // both buffers
        int ch = (int)std::min<size_t>(16, ((size_t)96 << 10) / per_col);
        ch = std::min(ch, (int)nb);
        const bool chunked = chunked_on && ch >= 1;
        const size_t lds_level = sizeof(val_t) * (size_t)nb + 2 * (size_t)ch * nb * (sizeof(val_t) + sizeof(u16)) + sizeof(u32) * ((size_t)nb + 2) + 16;
        const size_t lds_gather = 2 * sizeof(val_t) * (size_t)nb + sizeof(u32) * ((size_t)nb + 1);
        if (chunked)
        {
            static size_t allowed = 0;
            if (lds_level > allowed)
            {
                HIP_CHECK(hipFuncSetAttribute((const void *)block_trsv_level_chunked_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_level));
                HIP_CHECK(hipFuncSetAttribute((const void *)block_trsv_level_chunked_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_level));
                HIP_CHECK(hipFuncSetAttribute((const void *)block_trsv_gather_flat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(lds_gather, (size_t)1)));
                allowed = lds_level;
            }
        }
        for (size_t l = 0; l < (size_t)nlevel; l++)
        {
            const size_t n = (size_t)(level_ptr[l + 1] - level_ptr[l]), nbl = blk_level_ptr[l + 1] - blk_level_ptr[l];
            if (!n)
                continue;
            if (chunked)
            {
                if (nbl)
                    hipLaunchKernelGGL(block_trsv_gather_flat_kernel, dim3((unsigned)nbl), dim3(256), lds_gather, B.stream, d_blks + blk_level_ptr[l], (int)nb, d_x);
                if (upper)
                    hipLaunchKernelGGL(block_trsv_level_chunked_kernel<true>, dim3((unsigned)n), dim3(256), lds_level, B.stream, d_rows + level_ptr[l], (int)nb, d_x, ch);
                else
                    hipLaunchKernelGGL(block_trsv_level_chunked_kernel<false>, dim3((unsigned)n), dim3(256), lds_level, B.stream, d_rows + level_ptr[l], (int)nb, d_x, ch);
                continue;
            }
            if (nbl)
                hipLaunchKernelGGL(block_trsv_gather_kernel, dim3((unsigned)nbl), dim3(256), 0, B.stream, d_blks + blk_level_ptr[l], (int)nb, d_x);
            if (upper)
                hipLaunchKernelGGL(block_trsv_level_kernel<true>, dim3((unsigned)n), dim3(64), lds, B.stream, d_rows + level_ptr[l], (int)nb, d_x);
            else
                hipLaunchKernelGGL(block_trsv_level_kernel<false>, dim3((unsigned)n), dim3(64), lds, B.stream, d_rows + level_ptr[l], (int)nb, d_x);
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(x, d_x, sizeof(val_t) * (size_t)xlen, hipMemcpyDeviceToHost, B.stream));
        HIP_CHECK(hipStreamSynchronize(B.stream));
        HIP_CHECK(hipFree(d_rows));
        HIP_CHECK(hipFree(d_blks));
        HIP_CHECK(hipFree(d_x));
    }

    void pangulu_platform_0201001_block_spmv_add(pangulu_inblock_idx nb, pangulu_uint64_t nblk, pangulu_storage_slot_t *const *slots,
                                                 const pangulu_exblock_idx *src_seg, const pangulu_exblock_idx *dst_seg, const int *csr,
                                                 const calculate_type *x, calculate_type *y, pangulu_uint64_t xlen)
    {
        ensure_ready();
        std::lock_guard<std::mutex> g(B.mutex);
        flush_pending_getrf();
        HIP_CHECK(hipSetDevice(B.device));
        join_records(B.stream); // the sparse records of finished blocks are written on the records stream
        join_background(B.stream);
        std::vector<SpmvBlkD> hb(std::max<size_t>((size_t)nblk, 1));
        for (size_t i = 0; i < (size_t)nblk; i++)
        {
            const slot_t *s = slots[i];
            hb[i].ptr = csr[i] ? s->d_rowpointer : s->d_columnpointer;
            hb[i].idx = csr[i] ? s->d_columnindex : s->d_rowindex;
            hb[i].val = s->d_value;
            hb[i].src = src_seg[i];
            hb[i].dst = dst_seg[i];
            hb[i].csr = csr[i] ? 1u : 0u;
            hb[i].pad_ = 0;
        }
        SpmvBlkD *d_blks = nullptr;
        val_t *d_x = nullptr, *d_y = nullptr;
        HIP_CHECK(hipMalloc((void **)&d_blks, sizeof(SpmvBlkD) * hb.size()));
        HIP_CHECK(hipMalloc((void **)&d_x, sizeof(val_t) * (size_t)xlen));
        HIP_CHECK(hipMalloc((void **)&d_y, sizeof(val_t) * (size_t)xlen));
        HIP_CHECK(hipMemcpyAsync(d_blks, hb.data(), sizeof(SpmvBlkD) * hb.size(), hipMemcpyHostToDevice, B.stream));
        HIP_CHECK(hipMemcpyAsync(d_x, x, sizeof(val_t) * (size_t)xlen, hipMemcpyHostToDevice, B.stream));
        HIP_CHECK(hipMemcpyAsync(d_y, y, sizeof(val_t) * (size_t)xlen, hipMemcpyHostToDevice, B.stream));
        if (nblk)
            hipLaunchKernelGGL(block_spmv_add_kernel, dim3((unsigned)nblk), dim3(256), 0, B.stream, d_blks, (int)nb, d_x, d_y);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(y, d_y, sizeof(val_t) * (size_t)xlen, hipMemcpyDeviceToHost, B.stream));
        HIP_CHECK(hipStreamSynchronize(B.stream));
        HIP_CHECK(hipFree(d_blks));
        HIP_CHECK(hipFree(d_x));
        HIP_CHECK(hipFree(d_y));
    }

    // everything a recorded schedule depends on besides the block pattern
    static unsigned long long options_signature()
    {
        const long long v[] = {B.opt_host_mirror, B.opt_dense_permille, B.opt_profile, B.opt_assume_independent, B.opt_getrf_strict, B.opt_count_flops,
                               B.opt_group_chunk, B.opt_small_launch_tasks, B.opt_trsm_dense_permille, B.opt_two_streams, B.opt_records_stream,
                               B.opt_background_updates, B.opt_front_stages, B.opt_front_unit, B.opt_tiles_stages, B.opt_tiles_unit, B.opt_front_min_wgs};
        unsigned long long h = 1469598103934665603ull;
        for (long long x : v)
        {
            h ^= (unsigned long long)x;
            h *= 1099511628211ull;
        }
        return h;
    }

    static void drop_schedule()
    {
        for (Recorder::Seg &sg : REC.segs)
        {
            if (sg.h)
                (void)hipHostFree(sg.h);
            if (sg.twin)
                (void)hipFree(sg.twin);
        }
        REC = Recorder();
    }

    // host-side counters a recording accounts for (everything in B.stats that the launch code, not the device, fills)
    static void host_counters_delta(const pangulu_hip_stats_t &before, const pangulu_hip_stats_t &after, pangulu_hip_stats_t &d)
    {
        memset(&d, 0, sizeof(d));
        for (int c = 0; c < PANGULU_HIP_STAT_CLASSES; c++)
        {
            d.launches[c] = after.launches[c] - before.launches[c];
            d.tasks[c] = after.tasks[c] - before.tasks[c];
            d.alg_bytes[c] = after.alg_bytes[c] - before.alg_bytes[c];
        }
        d.trsm_dense_tasks = after.trsm_dense_tasks - before.trsm_dense_tasks;
    }
    static void host_counters_add(pangulu_hip_stats_t &to, const pangulu_hip_stats_t &d)
    {
        for (int c = 0; c < PANGULU_HIP_STAT_CLASSES; c++)
        {
            to.launches[c] += d.launches[c];
            to.tasks[c] += d.tasks[c];
            to.alg_bytes[c] += d.alg_bytes[c];
        }
        to.trsm_dense_tasks += d.trsm_dense_tasks;
    }

    // Static schedule of a factorisation (see Recorder).  cmd 1: start recording for `owner` (an opaque token: the handle);
    // 4: like 1, but record only -- nothing is launched (the scheduler's dry run at pangulu_init); 2: stop, the list is complete;
    // 3: replay the list if it belongs to `owner`, the options are those it was recorded under and none of the back-end's shared
    // resources the closures point into (GETRF scratch, mirror pool, progress words) has been freed or re-assigned since
    // (returns 0 when it was replayed, 1 when there is nothing valid to replay); 0: drop it (the owner's blocks are going
    // away).  Returns the number of recorded operations for cmd 2.  Not recorded (returns -1 on cmd 1 / 4): per-launch
    // profiling and the eager host mirror, whose copies and event pairs are not part of the list.
    long long pangulu_platform_0201001_schedule(int cmd, const void *owner)
    {
        ensure_ready();
        std::lock_guard<std::mutex> g(B.mutex);
        flush_pending_getrf();
        HIP_CHECK(hipSetDevice(B.device));
        switch (cmd)
        {
        case 0:
            if (!owner || owner == REC.owner)
            {
                HIP_CHECK(hipDeviceSynchronize());
                drop_schedule();
            }
            return 0;
        case 1:
        case 4:
            HIP_CHECK(hipDeviceSynchronize());
            drop_schedule();
            if (B.opt_profile || B.opt_host_mirror || !B.opt_assume_independent)
                return -1;
            REC.mode = cmd == 4 ? 2 : 1;
            REC.owner = owner;
            REC.signature = options_signature();
            REC.stats_before = B.stats;
            REC.wgs_before[0] = B.front_workgroups;
            REC.wgs_before[1] = B.general_workgroups;
            REC.wgs_before[2] = B.chase_launches;
            REC.wgs_before[3] = B.chase_solves;
            return 0;
        case 2:
        {
            if (REC.mode == 0)
                return -1;
            const int mode = REC.mode;
            REC.mode = 0;
            // the recorded run's kernels have read the segments in place; the replays read the HBM twins
            HIP_CHECK(hipDeviceSynchronize());
            for (Recorder::Seg &sg : REC.segs)
            {
                HIP_CHECK(hipMemcpy(sg.twin, sg.h, sg.cap, hipMemcpyHostToDevice));
                HIP_CHECK(hipHostFree(sg.h));
                sg.h = sg.d = nullptr;
            }
            host_counters_delta(REC.stats_before, B.stats, REC.stats_delta);
            REC.wgs_delta[0] = B.front_workgroups - REC.wgs_before[0];
            REC.wgs_delta[1] = B.general_workgroups - REC.wgs_before[1];
            REC.wgs_delta[2] = B.chase_launches - REC.wgs_before[2];
            REC.wgs_delta[3] = B.chase_solves - REC.wgs_before[3];
            if (mode == 2)
            {
                // a dry run launched nothing: the live counters go back to where they were (elapsed times and device-side
                // flop counts were not touched by it)
                for (int c = 0; c < PANGULU_HIP_STAT_CLASSES; c++)
                {
                    B.stats.launches[c] = REC.stats_before.launches[c];
                    B.stats.tasks[c] = REC.stats_before.tasks[c];
                    B.stats.alg_bytes[c] = REC.stats_before.alg_bytes[c];
                }
                B.stats.trsm_dense_tasks = REC.stats_before.trsm_dense_tasks;
                B.front_workgroups = REC.wgs_before[0];
                B.general_workgroups = REC.wgs_before[1];
                B.chase_launches = REC.wgs_before[2];
                B.chase_solves = REC.wgs_before[3];
            }
            REC.nb = B.nb_cfg;
            REC.generation = B.generation; // (the allocations of the recording itself are behind us)
            REC.valid = true;
            return (long long)REC.ops.size();
        }
        case 3:
            if (!REC.valid || REC.owner != owner || REC.signature != options_signature() || REC.generation != B.generation)
                return 1;
            for (auto &op : REC.ops)
                op();
            HIP_CHECK(hipGetLastError());
            host_counters_add(B.stats, REC.stats_delta);
            B.front_workgroups += REC.wgs_delta[0];
            B.general_workgroups += REC.wgs_delta[1];
            B.chase_launches += REC.wgs_delta[2];
            B.chase_solves += REC.wgs_delta[3];
            // (the records stream and the background stream may hold work the main stream has not joined: as after a real run)
            B.rec_dirty.store(true, std::memory_order_release);
            return 0;
        default:
            return -1;
        }
    }

    void *pangulu_platform_0201001_get_stream(void)
    {
        ensure_ready();
        return (void *)B.stream;
    }

    void pangulu_platform_0201001_get_stats(pangulu_hip_stats_t *out, int reset)
    {
        flush_pending_getrf_locked();
        if (reset && getenv("PANGULU_HIP_HOST_TIMING"))
        {
            fprintf(stderr, "[PanguLU-AMD] host seconds in the back-end: calls %.4f (ssssm %.4f, trsm %.4f, getrf %.4f, mirror jobs %.4f, staging waits %.4f)\n",
                    g_host_seconds[5], g_host_seconds[0], g_host_seconds[1], g_host_seconds[2], g_host_seconds[3], g_host_seconds[4]);
            for (double &x : g_host_seconds)
                x = 0;
        }
        ensure_ready();
        std::lock_guard<std::mutex> g(B.mutex);
        join_records(B.stream);
        join_background(B.stream);
        HIP_CHECK(hipStreamSynchronize(B.stream));
        harvest_events();
        unsigned long long f[16];
        HIP_CHECK(hipMemcpy(f, B.d_flops, sizeof(f), hipMemcpyDeviceToHost));
        if (getenv("PANGULU_HIP_DEBUG_SSSSM"))
            fprintf(stderr, "[ssssm_dense stamps, every 64th workgroup, shader clocks] bookkeeping+first step %llu | barrier A %llu | LDS stage (waits for the slab) %llu | barrier B %llu | next step + loads issued %llu | mfma %llu | C update %llu | slab steps %llu, empty workgroups %llu\n",
                    f[8], f[9], f[10], f[11], f[12], f[13], f[14], f[15] & 0xFFFFFFFFull, f[15] >> 32);
        if (getenv("PANGULU_HIP_DEBUG_TRSM"))
            fprintf(stderr, "[trsm stamps, every 64th workgroup, shader clocks] setup+x loads %llu | prefetch issue %llu | barrier A %llu | stage %llu | barrier B %llu | mfma loop %llu | tail chain %llu | stores %llu, empty workgroups %llu\n",
                    f[8], f[9], f[10], f[11], f[12], f[13], f[14], f[15] & ((1ull << 40) - 1), f[15] >> 40);
        if (getenv("PANGULU_HIP_DEBUG_GETRF"))
            fprintf(stderr, "[getrf stamps, block 0, shader clocks; tiled kernel: 1 = diag+priority tiles, 2 = trailing passes, 3 = wait for LU, 4 = substitution (wavefront 0), lu = tile LU (wavefront 7)] prologue %llu | 1 %llu | 2 %llu | 3 %llu | 4 %llu | loop %llu | gather %llu | lu %llu\n",
                    f[8], f[9], f[10], f[11], f[12], f[13], f[14], f[15]);
        for (int c = 1; c <= 5; c++)
            B.stats.flops[c] = (double)f[c];
        B.stats.mfma_flops_executed = 8192.0 * (double)f[6]; // 16 x 16 x 16 products counted by the MFMA update kernels
        B.stats.ssssm_front_workgroups = B.front_workgroups;
        B.stats.ssssm_general_workgroups = B.general_workgroups;
        B.stats.chase_launches = B.chase_launches;
        B.stats.chase_solves = B.chase_solves;
        if (out)
            *out = B.stats;
        if (reset)
        {
            memset(&B.stats, 0, sizeof(B.stats));
            B.mfma_flops_executed = 0;
            B.front_workgroups = B.general_workgroups = 0;
            B.chase_launches = B.chase_solves = 0;
            HIP_CHECK(hipMemset(B.d_flops, 0, sizeof(f)));
        }
    }
}
