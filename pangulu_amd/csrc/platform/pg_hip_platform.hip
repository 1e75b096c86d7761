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
#include "pg_hip_trsm_ring.h"
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


#if defined(PG_DENSE_PANELS)
#include "pg_hip_getrf_tiled.h"
#include "pg_hip_getrf_pipe.h"

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

#include "pg_hip_solve_ops.h"

// =================================================================================================================
// host side of the back-end
// =================================================================================================================
namespace
{

#include "pg_hip_block_solve.h"

#include "pg_hip_backend.h"

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

#include "pg_hip_launch_ssssm.h"

#include "pg_hip_launch_trsm.h"

#include "pg_hip_launch_getrf.h"

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
        {
            std::lock_guard<std::mutex> g(B.mutex); // (the joins touch back-end state: see memcpy below)
            flush_pending_getrf();
            join_records(B.stream);
            join_background(B.stream);
        }
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
        // ordered after everything queued on the back-end stream, complete on return.
        // The host-staged transport calls this from the scheduler's thread (pg_comm_socket.h, isend_block / recv_block) while the
        // launcher thread is inside hybrid_batched: the joins clear back-end state (bg_tiles, the record flag), so they and the
        // enqueue happen under the back-end's mutex (round 6: eight ranks on kkt(64) over the host transport died inside
        // bg_tiles' hash table within seconds, profiles/r06zo_*); the wait is for THIS copy only, outside the mutex.
        hipEvent_t done;
        HIP_CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        {
            std::lock_guard<std::mutex> g(B.mutex);
            flush_pending_getrf();
            join_records(B.stream);
            join_background(B.stream);
            HIP_CHECK(hipMemcpyAsync(dst, src, count, kind_of(kind), B.stream));
            HIP_CHECK(hipEventRecord(done, B.stream));
        }
        HIP_CHECK(hipEventSynchronize(done));
        HIP_CHECK(hipEventDestroy(done));
    }

    void pangulu_platform_0201001_memcpy_async(void *dst, const void *src, size_t count, unsigned int kind, void *stream)
    {
        ensure_ready();
        // stream == NULL is what the reference host passes from its receive thread
        // (src/pangulu_communication.c:1850,1880): use the back-end stream so later kernels are ordered behind it
        hipStream_t s = stream ? (hipStream_t)stream : B.stream;
        hipEvent_t done = nullptr;
        if (!stream)
            HIP_CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        {
            std::lock_guard<std::mutex> g(B.mutex); // (a receive thread beside the launcher: as in memcpy above)
            flush_pending_getrf();
            if (kind != 0)
            {
                join_records(s); // (uploads of received blocks write receive slots, which no sparsify job touches)
                join_background(s);
            }
            HIP_CHECK(hipMemcpyAsync(dst, src, count, kind_of(kind), s));
            if (done)
                HIP_CHECK(hipEventRecord(done, s));
        }
        if (done)
        {
            HIP_CHECK(hipEventSynchronize(done)); // the source is pageable host memory the caller may reuse at once
            HIP_CHECK(hipEventDestroy(done));
        }
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
        case PANGULU_HIP_OPT_QUERY_FREE_MIB:
        {
            ensure_ready();
            size_t free_b = 0, total_b = 0;
            HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
            return (int)(free_b >> 20);
        }
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
        for (hipEvent_t e : REC.early_events)
            (void)hipEventDestroy(e);
#if defined(PG_DENSE_UPDATES)
        MP.early.clear();
        MP.early_h = MP.early_d = nullptr; // (the segment went with the recording)
        MP.early_cap = MP.early_used = 0;
#endif
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
        // pure queries first: they launch nothing and must not flush a held GETRF (the scheduler's log asks for the
        // operation count behind every platform call of a recording run -- with the flush in front, every such query ended the
        // GETRF / solve chase of the recorded run and froze a chase-free launch list into every replay)
        if (cmd == 6) // operations recorded so far (while recording) / in the list; a held GETRF's launches join the next entry's range
            return (long long)REC.ops.size();
        if (cmd == 9) // would cmd 7 / cmd 3 replay?  (validity rule only: owner, options signature, generation of the shared resources)
            return (REC.valid && REC.owner == owner && REC.signature == options_signature() && REC.generation == B.generation) ? 0 : 1;
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
        case 5: // (5: like 1, with the descriptor segments packed -- a multi-rank run's thousands of small batches)
            HIP_CHECK(hipDeviceSynchronize());
            drop_schedule();
            if (B.opt_profile || B.opt_host_mirror || !B.opt_assume_independent)
                return -1;
            REC.mode = cmd == 4 ? 2 : 1;
            REC.pack = true; // (round 6: every recording packs its descriptor segments, see acquire_segment)
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
#if defined(PG_DENSE_UPDATES)
            if (!MP.early.empty())
            {
                // (every launch function flushes what it moved: this cannot happen)
                fprintf(stderr, "[PanguLU-AMD ERROR] %zu early densify jobs were never given a wait point\n", MP.early.size());
                exit(EXIT_FAILURE);
            }
            if (getenv("PANGULU_AMD_TRACE") && MP.early_chunks)
                fprintf(stderr, "[pangulu_amd trace] schedule: %llu first-touch densify jobs moved into %llu prologue chunks on the early stream\n",
                        MP.early_jobs, MP.early_chunks);
            MP.early_jobs = MP.early_chunks = 0;
#endif
            REC.nb = B.nb_cfg;
            REC.generation = B.generation; // (the allocations of the recording itself are behind us)
            REC.valid = true;
            return (long long)REC.ops.size();
        }
        case 7: // a replay in RANGES begins (pangulu_platform_0201001_schedule_range): same validity rule as 3; the prologue goes out
            if (!REC.valid || REC.owner != owner || REC.signature != options_signature() || REC.generation != B.generation)
                return 1;
            for (auto &op : REC.prologue)
                op();
            return 0;
        case 8: // ... and ends: counters of one factorisation, streams as after a real run
            if (!REC.valid || REC.owner != owner)
                return 1;
            HIP_CHECK(hipGetLastError());
            host_counters_add(B.stats, REC.stats_delta);
            B.front_workgroups += REC.wgs_delta[0];
            B.general_workgroups += REC.wgs_delta[1];
            B.chase_launches += REC.wgs_delta[2];
            B.chase_solves += REC.wgs_delta[3];
            B.rec_dirty.store(true, std::memory_order_release);
            return 0;
        case 3:
            if (!REC.valid || REC.owner != owner || REC.signature != options_signature() || REC.generation != B.generation)
                return 1;
            for (auto &op : REC.prologue) // (early densify chunks, on their own stream: the main list waits for their events)
                op();
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

    // Replay of the operations [first, last) of the owner's recorded list (between cmd 7 and cmd 8 of schedule): the multi-rank
    // host replays its own log batch by batch, waiting for the blocks of other ranks in between.  Returns 0, or 1 when there is
    // nothing valid to replay.
    int pangulu_platform_0201001_schedule_range(const void *owner, long long first, long long last)
    {
        std::lock_guard<std::mutex> g(B.mutex);
        if (!REC.valid || REC.owner != owner || first < 0 || last > (long long)REC.ops.size() || first > last)
            return 1;
        HIP_CHECK(hipSetDevice(B.device));
        for (long long i = first; i < last; i++)
            REC.ops[(size_t)i]();
        return 0;
    }

    // A marker at the current point of the main stream and nothing else: in a ranged replay the joins a marker needs (records
    // stream, background stream) are operations of the recorded list already -- marker_record appended them while recording.
    void *pangulu_platform_0201001_marker_record_replay(void)
    {
        std::lock_guard<std::mutex> g(B.mutex);
        HIP_CHECK(hipSetDevice(B.device));
        static std::vector<hipEvent_t> ring;
        static size_t next = 0;
        if (ring.empty())
        {
            ring.resize(4096);
            for (hipEvent_t &e : ring)
                HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        hipEvent_t e = ring[next];
        next = (next + 1) % ring.size();
        HIP_CHECK(hipEventRecord(e, B.stream));
        return (void *)e;
    }

    void *pangulu_platform_0201001_get_stream(void)
    {
        ensure_ready();
        return (void *)B.stream;
    }

    void pangulu_platform_0201001_get_memory(unsigned long long out[4])
    {
        std::lock_guard<std::mutex> g(B.mutex);
        out[0] = out[1] = out[2] = out[3] = 0;
#if defined(PG_DENSE_UPDATES)
        out[0] = (unsigned long long)MP.chunks.size() * MP.chunk_bytes;
        out[3] = (unsigned long long)MP.peak;
#endif
        out[1] = (unsigned long long)REC.descriptor_bytes;
        out[2] = B.getrf_scratch ? (unsigned long long)B.getrf_scratch_slots * std::max(sizeof(val_t), sizeof(double)) * (size_t)B.nb_cfg * B.nb_cfg : 0ull;
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
        unsigned long long f[PG_FLOP_WORDS];
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
        if (getenv("PANGULU_HIP_DEBUG_GETRF"))
            fprintf(stderr, "[getrf_pipe stamps, block 0, shader clocks; factorisation wavefront: wait A %llu | tile LU %llu | inverses %llu | wait B %llu | wait C %llu; trailing wavefront 1: update %llu | wait B %llu | panel+strip %llu | wait C %llu | hand-over %llu | wait A %llu]\n",
                    f[8 + 16], f[8 + 17], f[8 + 18], f[8 + 19], f[8 + 20], f[8 + 24], f[8 + 25], f[8 + 26], f[8 + 27], f[8 + 28], f[8 + 29]);
        for (int c = 1; c <= 5; c++)
            B.stats.flops[c] = (double)f[c];
        B.stats.mfma_flops_executed = 8192.0 * (double)(f[6] + f[7]); // 16 x 16 x 16 products counted by the MFMA update kernels
        B.stats.ssssm_front_flops_executed = 8192.0 * (double)f[7];    // ... the dense-front kernel's share
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
