// Stand-alone check + timing of the update kernels on a dense front: ssssm_front_f64_kernel<STAGES> (pg_hip_front.h) against
// ssssm_dense_f64_kernel (pg_hip_dense.h) with every tile live.  The shape of one level of a dense separator: P x P
// destination blocks C(i,j) -= sum_q A_q(i) B_q(j), nb = 256, four 128 x 128 tiles each.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -o front_gemm.bin front_gemm.hip
//   ./front_gemm.bin [P = 40] [Q = 1]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef double val_t;
typedef unsigned int u32;
typedef unsigned short u16;
#define PG_PLANES 1
struct BlkView { const u32 *ptr; const u16 *idx; val_t *val; };
struct SsssmTaskD { BlkView a, b; double sign; u32 count; u32 has_map; unsigned short amap[16], bmap_t[16]; }; // = pg_hip_platform.hip
static_assert(sizeof(SsssmTaskD) == 128, "descriptor layout");
struct SsssmWorkD { val_t *cdense; u32 task_begin, task_end; u32 atomic, slab_mask; u32 tile, pad_; };
__device__ inline u32 ptr0(const u32 *p, int i) { return i == 0 ? 0u : p[i]; }
__device__ inline unsigned long long wave_sum(unsigned long long v) { for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64); return v; }
typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double real_t;
__host__ __device__ inline val_t v_make(real_t r) { return r; }
__constant__ int c_xcd_swizzle = 1;
__device__ inline unsigned logical_block_id(unsigned per_unit)
{
    const unsigned n = gridDim.x, b = blockIdx.x;
    const unsigned round = 8u * per_unit, full = (n / round) * round;
    if (!c_xcd_swizzle || b >= full)
        return b;
    const unsigned x = b & 7, idx = b >> 3;
    return (x + 8u * (idx / per_unit)) * per_unit + idx % per_unit;
}
#ifdef TL_PROBE
// cycle probes inside ssssm_tiles_f64_kernel (wavefronts 0 and 5 of every workgroup): where does a slab step spend its time?
__device__ unsigned long long g_probe[8];
#define TL_PROBE_DECL unsigned long long pr_t = __builtin_readcyclecounter(), pr_sum[7] = {0, 0, 0, 0, 0, 0, 0};
#define TL_MARK(i) { const unsigned long long n_ = __builtin_readcyclecounter(); pr_sum[i] += n_ - pr_t; pr_t = n_; }
#define TL_PROBE_FLUSH if (lane == 0 && (wave == 0 || wave == 5)) { for (int i_ = 0; i_ < 7; i_++) atomicAdd(&g_probe[i_], pr_sum[i_]); }
#define TL_PROBE_STEP pr_sum[5]++;
// ... and the life of a work item outside the step loop (wavefront 0 of every workgroup): TL_ITEM(i) adds the cycles since the previous
// stamp to phase i; phase 15 counts the items
__device__ unsigned long long g_item[16];
#define TL_ITEM_DECL unsigned long long it_t = __builtin_readcyclecounter();
#define TL_ITEM(i) { const unsigned long long n_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g_item[i], n_ - it_t); it_t = n_; }
#define TL_ITEM_COUNT if (threadIdx.x == 0) atomicAdd(&g_item[15], 1ull);
#endif
#include "../../pangulu_amd/csrc/platform/pg_hip_dense.h"
#include "../../pangulu_amd/csrc/platform/pg_hip_front.h"
#include "../experiments/ssssm_tiles.h"
#include "../experiments/pg_hip_pieces.h"
#ifdef SW_PROBE
__device__ unsigned long long g_sw_probe[8];
#endif
#include "../experiments/pg_hip_stream.h"
#include "../experiments/front_k32.h"
#include "../experiments/front_n64.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Problem
{
    int nb = 256, P = 0, Q = 0;
    size_t mb = 0; // doubles per mirror
    std::vector<double> h; // host copy of all mirrors: [A: Q*P][B: Q*P][C: P*P]
    std::vector<unsigned short> maps; // 16 words per mirror
    double products = 0; // live 16 x 16 x 16 products of one launch
    double *d = nullptr;
    SsssmTaskD *dT = nullptr;
    SsssmWorkD *dW = nullptr;
    size_t nwork = 0;
    // the stream kernel (pg_hip_stream.h): work items that carry their first step slot, the builder's scratch
    SsssmWorkD *dWs = nullptr;
    SsssmItemInfoD *dInfo = nullptr;
    SsssmStepD *dSteps = nullptr;
    double *A(int q, int i) { return d + (size_t)(q * P + i) * mb; }
    double *Bm(int q, int j) { return d + (size_t)(Q * P + q * P + j) * mb; }
    double *C(int i, int j) { return d + (size_t)(2 * Q * P + i * P + j) * mb; }
    size_t offA(int q, int i) { return (size_t)(q * P + i) * mb; }
    size_t offB(int q, int j) { return (size_t)(Q * P + q * P + j) * mb; }
    size_t offC(int i, int j) { return (size_t)(2 * Q * P + i * P + j) * mb; }
};

// fill < 100: every operand gets a random occupancy map made of contiguous ranges of live 16-row pieces per 16-column slab
// (what fill-in looks like), about `fill` percent of the pieces live; the values of dead pieces stay random garbage -- a kernel
// that multiplies them fails the check
static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static unsigned rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (unsigned)(rng_state >> 33); }
// fill = -k: every slab live with exactly k contiguous 16-row pieces in each 128-row half (`transposed`: the map of a B operand,
// k live column pieces per half in every row slab) -- every slab step of every tile then has k x k live products
void random_map(unsigned short *map, int fill, bool transposed = false)
{
    if (fill < 0)
    {
        const int k = -fill;
        unsigned short w[16];
        for (int c = 0; c < 16; c++)
        {
            const unsigned lo = ((1u << k) - 1u) << (rnd() % (unsigned)(9 - k)), hi = ((1u << k) - 1u) << (rnd() % (unsigned)(9 - k));
            w[c] = (unsigned short)(lo | (hi << 8));
        }
        for (int c = 0; c < 16; c++)
        {
            if (!transposed) { map[c] = w[c]; continue; }
            map[c] = 0;
            for (int r = 0; r < 16; r++) map[c] |= (unsigned short)(((w[r] >> c) & 1u) << r);
        }
        return;
    }
    for (int c = 0; c < 16; c++)
    {
        if (fill >= 100) { map[c] = 0xFFFF; continue; }
        if ((int)(rnd() % 100) >= fill + (100 - fill) / 2) { map[c] = 0; continue; } // some slabs are empty altogether
        int len = 1 + (int)(rnd() % (unsigned)(1 + (32 * fill) / 100));
        if (len > 16) len = 16;
        const int r0 = (int)(rnd() % (unsigned)(17 - len));
        map[c] = (unsigned short)(((1u << len) - 1u) << r0);
    }
}

void build(Problem &X, int P, int Q, bool keep_host, int fill = 100)
{
    X.P = P; X.Q = Q;
    X.mb = mirror_plane_stride(X.nb);
    const size_t nm = (size_t)2 * Q * P + (size_t)P * P;
    CK(hipMalloc(&X.d, sizeof(double) * X.mb * nm));
    std::vector<double> one(X.mb);
    if (keep_host) X.h.resize(X.mb * nm);
    unsigned long long s = 88172645463325252ull;
    for (size_t m = 0; m < nm; m++)
    {
        for (size_t i = 0; i < (size_t)X.nb * X.nb; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; one[i] = (double)(s >> 11) / 9007199254740992.0 - 0.5; }
        unsigned short *map = reinterpret_cast<unsigned short *>(one.data() + (size_t)X.nb * X.nb);
        random_map(map, m < (size_t)2 * Q * P ? fill : 100, m >= (size_t)Q * P);
        X.maps.insert(X.maps.end(), map, map + 16);
        CK(hipMemcpy(X.d + m * X.mb, one.data(), sizeof(double) * X.mb, hipMemcpyHostToDevice));
        if (keep_host) memcpy(X.h.data() + m * X.mb, one.data(), sizeof(double) * X.mb);
    }
    std::vector<SsssmTaskD> T((size_t)P * P * Q);
    std::vector<SsssmWorkD> W((size_t)P * P * 4);
    for (int i = 0; i < P; i++)
        for (int j = 0; j < P; j++)
        {
            const size_t g = (size_t)i * P + j;
            for (int q = 0; q < Q; q++)
            {
                SsssmTaskD &t = T[g * Q + q];
                memset(&t, 0, sizeof(t));
                t.a.val = X.A(q, i); t.b.val = X.Bm(q, j); t.sign = 1.0; t.count = 1; t.has_map = 1;
                const unsigned short *ma = X.maps.data() + 16 * (size_t)(q * P + i), *mbm = X.maps.data() + 16 * (size_t)(Q * P + q * P + j);
                for (int c = 0; c < 16; c++) { t.amap[c] = ma[c]; t.bmap_t[c] = 0; }
                for (int c = 0; c < 16; c++) for (int r = 0; r < 16; r++) if ((mbm[c] >> r) & 1) t.bmap_t[r] |= (unsigned short)(1u << c); // word r: column slabs live in row slab r
                for (int sl = 0; sl < 16; sl++) X.products += (double)__builtin_popcount(t.amap[sl]) * (double)__builtin_popcount(t.bmap_t[sl]);
            }
            for (int tl = 0; tl < 4; tl++)
                W[g * 4 + tl] = SsssmWorkD{X.C(i, j), (u32)(g * Q), (u32)(g * Q + Q), 0u, 0u, (u32)tl, 0u};
        }
    X.nwork = W.size();
    CK(hipMalloc(&X.dT, sizeof(SsssmTaskD) * T.size()));
    CK(hipMalloc(&X.dW, sizeof(SsssmWorkD) * W.size()));
    CK(hipMemcpy(X.dT, T.data(), sizeof(SsssmTaskD) * T.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(X.dW, W.data(), sizeof(SsssmWorkD) * W.size(), hipMemcpyHostToDevice));
    // stream kernel: slot bound per item = Q tasks x 16 slabs
    for (size_t w = 0; w < W.size(); w++)
        W[w].pad_ = (u32)((w * (size_t)Q * 16) << 1);
    CK(hipMalloc(&X.dWs, sizeof(SsssmWorkD) * W.size()));
    CK(hipMemcpy(X.dWs, W.data(), sizeof(SsssmWorkD) * W.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&X.dInfo, sizeof(SsssmItemInfoD) * W.size()));
    CK(hipMalloc(&X.dSteps, sizeof(SsssmStepD) * W.size() * (size_t)Q * 16));
}

// which: 0 = general kernel; otherwise 100 * unit_destinations + 10 * prefetch + stages
void launch(int which, Problem &X)
{
    static const unsigned cap = getenv("GRID_CAP") ? (unsigned)atoi(getenv("GRID_CAP")) : 0u; // (timing experiments: only the first workgroups)
    const unsigned grid = cap && cap < X.nwork ? cap : (unsigned)X.nwork;
    unsigned long long *none = nullptr;
    if (which == 0)
    {
        hipLaunchKernelGGL(ssssm_dense_f64_kernel, dim3(grid), dim3(DG_THREADS), 0, 0, X.dT, X.nb, none, none, X.dW);
        CK(hipGetLastError());
        return;
    }
    if (which >= 60000)
    {
        // round 6: persistent sixteen-wavefront workgroups walking a stream of slab steps (pg_hip_stream.h); which % 1000 = workgroups (0: 256)
        static bool allowed = false;
        if (!allowed)
        {
            CK(hipFuncSetAttribute((const void *)ssssm_stream_f64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SW_LDS_BYTES));
            allowed = true;
        }
        unsigned wgs = (unsigned)(which % 1000) ? (unsigned)(which % 1000) : 256u;
        const unsigned need = (unsigned)((grid + 31) / 32 * 32);
        if (wgs > need) wgs = need;
        hipLaunchKernelGGL(ssssm_stream_build_kernel, dim3(grid), dim3(256), 0, 0, X.dT, X.nb, X.dWs, X.dInfo, X.dSteps);
        hipLaunchKernelGGL(ssssm_stream_f64_kernel, dim3(wgs), dim3(SW_THREADS), SW_LDS_BYTES, 0, X.nb, X.dWs, X.dInfo, X.dSteps, grid, none);
        CK(hipGetLastError());
        return;
    }
    if (which >= 50000)
    {
        // round 5: compacted, piece-indexed staging (pg_hip_pieces.h)
        hipLaunchKernelGGL(ssssm_tilesp_f64_kernel, dim3(grid), dim3(FR_THREADS), 0, 0, X.dT, X.nb, X.dW, none, 4u * (unsigned)((which / 100) % 100));
        CK(hipGetLastError());
        return;
    }
    if (which >= 40000)
    {
        // EXPERIMENT (tools/experiments/front_n64.h): 128 x 64 tiles, four wavefronts per workgroup, three workgroups per CU
        hipLaunchKernelGGL(ssssm_front_n64_f64_kernel, dim3(2 * grid), dim3(F64N_THREADS), 0, 0, X.dT, X.nb, X.dW, none, 4u * (unsigned)((which / 100) % 100));
        CK(hipGetLastError());
        return;
    }
    if (which >= 30000)
    {
        // EXPERIMENT (tools/experiments/front_k32.h): K = 32 per barrier, one workgroup of sixteen wavefronts per CU
        static bool allowed = false;
        const size_t lds_b = 2 * sizeof(double) * F32_STAGE_DOUBLES;
        if (!allowed)
        {
            CK(hipFuncSetAttribute((const void *)ssssm_front32_f64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
            allowed = true;
        }
        hipLaunchKernelGGL(ssssm_front32_f64_kernel, dim3(grid), dim3(F32_THREADS), lds_b, 0, X.dT, X.nb, X.dW, none, 4u * (unsigned)((which / 100) % 100));
        CK(hipGetLastError());
        return;
    }
    if (which >= 20000)
    {
        // (tools/experiments/ssssm_lds_counters_for_barriers.patch adds a variant `which % 10 == 3`)
        hipLaunchKernelGGL(ssssm_tilesv_f64_kernel, dim3(grid), dim3(FR_THREADS), 0, 0, X.dT, X.nb, X.dW, none, 4u * (unsigned)((which / 100) % 100));
        CK(hipGetLastError());
        return;
    }
    if (which >= 10000)
    {
        const unsigned unit = 4u * (unsigned)((which / 100) % 100);
        const int st = which % 10;
        if (st == 2) hipLaunchKernelGGL((ssssm_tiles_f64_kernel<2>), dim3(grid), dim3(FR_THREADS), 0, 0, X.dT, X.nb, X.dW, none, unit);
        else if (st == 3) hipLaunchKernelGGL((ssssm_tiles_f64_kernel<3>), dim3(grid), dim3(FR_THREADS), 0, 0, X.dT, X.nb, X.dW, none, unit);
        else hipLaunchKernelGGL((ssssm_tiles_f64_kernel<4>), dim3(grid), dim3(FR_THREADS), 0, 0, X.dT, X.nb, X.dW, none, unit);
        CK(hipGetLastError());
        return;
    }
    const unsigned unit = 4u * (unsigned)(which / 100);
    const int pf = (which / 10) % 10, st = which % 10;
#define GO(S_, P_) hipLaunchKernelGGL((ssssm_front_f64_kernel<S_, P_>), dim3(grid), dim3(FR_THREADS), 0, 0, X.dT, X.nb, X.dW, none, unit)
    if (st == 2 && !pf) GO(2, false);
    else if (st == 2) GO(2, true);
    else if (st == 3 && !pf) GO(3, false);
    else if (st == 3) GO(3, true);
    else if (st == 4 && !pf) GO(4, false);
    else GO(4, true);
#undef GO
    CK(hipGetLastError());
}

static char name_buf[128];
const char *name_of(int which)
{
    if (which == 0)
        return "round-2 kernel (pg_hip_dense.h)";
    if (which >= 60000)
    {
        snprintf(name_buf, sizeof(name_buf), "stream kernel (16 wavefronts, 4 stages, persistent), %d workgroups", which % 1000 ? which % 1000 : 256);
        return name_buf;
    }
    if (which >= 50000)
    {
        snprintf(name_buf, sizeof(name_buf), "pieces kernel (ring of 32 piece slots, batches), XCD unit %d dest.", (which / 100) % 100);
        return name_buf;
    }
    if (which >= 40000)
    {
        snprintf(name_buf, sizeof(name_buf), "front kernel 128 x 64 tiles, 4 wavefronts, 3 workgroups / CU, unit %d dest.", (which / 100) % 100);
        return name_buf;
    }
    if (which >= 30000)
    {
        snprintf(name_buf, sizeof(name_buf), "front kernel K = 32 per barrier, 16 wavefronts, XCD unit %d dest.", (which / 100) % 100);
        return name_buf;
    }
    if (which >= 20000)
    {
        snprintf(name_buf, sizeof(name_buf), "tiles kernel, DMA issue behind the first products, XCD unit %d dest.", (which / 100) % 100);
        return name_buf;
    }
    if (which >= 10000)
    {
        snprintf(name_buf, sizeof(name_buf), "tiles kernel (DMA, strided pieces), %d LDS stages, XCD unit %d dest.", which % 10, (which / 100) % 100);
        return name_buf;
    }
    snprintf(name_buf, sizeof(name_buf), "front kernel, %d LDS stages, %s, XCD unit %d dest.", which % 10, (which / 10) % 10 ? "fragment prefetch" : "no prefetch  ", which / 100);
    return name_buf;
}

// C -= sum_q A_q B_q with the dead pieces of the operands taken as zero
void reference(Problem &X, int i, int j, std::vector<double> &ref)
{
    const int nb = X.nb;
    const double *c0 = X.h.data() + X.offC(i, j);
    for (size_t e = 0; e < (size_t)nb * nb; e++) ref[e] = c0[e];
    for (int q = 0; q < X.Q; q++)
    {
        const double *a = X.h.data() + X.offA(q, i), *b = X.h.data() + X.offB(q, j);
        const unsigned short *ma = X.maps.data() + 16 * (size_t)(q * X.P + i), *mbm = X.maps.data() + 16 * (size_t)(X.Q * X.P + q * X.P + j);
        for (int n = 0; n < nb; n++)
            for (int k = 0; k < nb; k++)
            {
                if (!((mbm[n >> 4] >> (k >> 4)) & 1)) continue; // B piece (row slab k, column slab n) dead
                const double bkn = b[(size_t)n * nb + k];
                for (int m = 0; m < nb; m++)
                    if ((ma[k >> 4] >> (m >> 4)) & 1)
                        ref[(size_t)n * nb + m] -= a[(size_t)k * nb + m] * bkn;
            }
    }
}

int main(int argc, char **argv)
{
    const int P = argc > 1 ? atoi(argv[1]) : 40, Q = argc > 2 ? atoi(argv[2]) : 1, fill = argc > 3 ? atoi(argv[3]) : 100;
    // ---- correctness on a small front: every entry of some destinations against a host product
    for (int cf : {100, 45})
    {
        Problem X;
        build(X, 3, 2, true, cf);
        const int nb = X.nb;
        std::vector<double> ref((size_t)nb * nb), got((size_t)X.mb);
        std::vector<int> kinds = cf == 100 ? std::vector<int>{0, 102, 113, 104, 10102, 10103, 10104, 20102, 30100, 40100, 50102, 60000, 60032} : std::vector<int>{0, 10102, 10103, 10104, 10802, 20102, 20802, 50102, 50802, 60000, 60032};
        for (int which : kinds)
        {
            for (int i = 0; i < X.P; i++)
                for (int j = 0; j < X.P; j++)
                    CK(hipMemcpy(X.C(i, j), X.h.data() + X.offC(i, j), sizeof(double) * X.mb, hipMemcpyHostToDevice));
            launch(which, X);
            CK(hipDeviceSynchronize());
            double worst = 0;
            for (int i = 0; i < X.P; i += 2)
                for (int j = 0; j < X.P; j += 2)
                {
                    reference(X, i, j, ref);
                    CK(hipMemcpy(got.data(), X.C(i, j), sizeof(double) * X.mb, hipMemcpyDeviceToHost));
                    for (size_t e = 0; e < (size_t)nb * nb; e++)
                        worst = fmax(worst, fabs(got[e] - ref[e]));
                }
            printf("check (fill %3d%%)  %-66s max |C - ref| = %.3e %s\n", cf, name_of(which), worst, worst < 1e-11 ? "ok" : "WRONG");
        }
        CK(hipFree(X.d)); CK(hipFree(X.dT)); CK(hipFree(X.dW)); CK(hipFree(X.dWs)); CK(hipFree(X.dInfo)); CK(hipFree(X.dSteps));
    }
    // ---- timing
    Problem X;
    build(X, P, Q, false, fill);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double flop = 8192.0 * X.products;
    printf("front %d x %d destinations of 256 x 256, %d update(s) queued on each, %d%% fill pattern: %zu workgroups, %.3f TFLOP of live 16x16x16 products per launch (%.1f%% of dense)\n", P, P, Q,
           fill, X.nwork, flop / 1e12, 100.0 * flop / (2.0 * 256 * 256 * 256 * (double)P * P * Q));
    std::vector<int> kinds = fill >= 100 ? std::vector<int>{112, 20102, 60000, 60224, 112, 20102, 60000, 60224} : std::vector<int>{20102, 20802, 60000, 60224, 20102, 20802, 60000, 60224};
#ifdef TL_PROBE
    kinds = {20102, 50102, 20102, 50102};
#endif
    for (int which : kinds)
    {
#ifdef TL_PROBE
        unsigned long long zero[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_probe), zero, 8 * sizeof(unsigned long long)));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_item), zero, sizeof(zero)));
#endif
        float best = 1e30f, sum = 0;
        for (int rep = 0; rep < 5; rep++)
        {
            CK(hipEventRecord(a));
            launch(which, X);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            best = fminf(best, ms);
            if (rep) sum += ms;
        }
#ifdef SW_PROBE
        if (which >= 60000)
        {
            unsigned long long pr[8];
            CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_sw_probe), sizeof(pr)));
            const double st = (double)pr[7];
            printf("probe  stream kernel, cycles per step and wavefront (%.0f steps sampled): wait for the slab %.0f, barrier %.0f, fragment reads + wait %.0f, first products %.0f, "
                   "request + queue %.0f, other three quarters + loop end %.0f: %.0f in all\n",
                   st, pr[0] / st, pr[1] / st, pr[2] / st, pr[3] / st, pr[4] / st, pr[5] / st, (pr[0] + pr[1] + pr[2] + pr[3] + pr[4] + pr[5]) / st);
            unsigned long long zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_sw_probe), zero8, sizeof(zero8)));
        }
#endif
        printf("time   %-66s best %8.3f ms = %6.2f TFLOP/s executed, mean of 4 %8.3f ms = %6.2f\n", name_of(which), best, flop / best / 1e9, sum / 4, flop / (sum / 4) / 1e9);
#ifdef TL_PROBE
        unsigned long long pr[8];
        CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_probe), sizeof(pr)));
        const double st = (double)pr[5];
        {
            unsigned long long it[16];
            CK(hipMemcpyFromSymbol(it, HIP_SYMBOL(g_item), sizeof(it)));
            const double ni = (double)it[15];
            if (ni > 0)
                printf("item   cycles per work item (%.0f items): work item + descriptors + candidates %.0f, scan / tables / barriers %.0f, batches + preload issue %.0f, first issue %.0f, "
                       "step loop %.0f, epilogue %.0f: %.0f in all; %.1f live steps per item\n",
                       ni, it[0] / ni, it[1] / ni, it[2] / ni, it[3] / ni, it[4] / ni, it[5] / ni, (it[0] + it[1] + it[2] + it[3] + it[4] + it[5]) / ni, st / 2.0 / ni);
        }
        if (which >= 50000)
            printf("probe  cycles per slab step and wavefront (%.0f steps sampled): wait for the batch %.0f, barrier %.0f, record + fragments + first quarter %.0f, issue of following batches %.0f, other three quarters %.0f, between batches %.0f: %.0f in all\n",
                   st, pr[0] / st, pr[1] / st, pr[2] / st, pr[3] / st, pr[4] / st, pr[6] / st, (pr[0] + pr[1] + pr[2] + pr[3] + pr[4] + pr[6]) / st);
        else if (which >= 20000)
            printf("probe  cycles per slab step and wavefront (%.0f steps sampled): record reads + wait for the slab %.0f, barrier %.0f, fragments + first quarter %.0f, issue of the next slab %.0f, other three quarters %.0f, between steps %.0f: %.0f in all\n",
                   st, pr[0] / st, pr[1] / st, pr[2] / st, pr[3] / st, pr[4] / st, pr[6] / st, (pr[0] + pr[1] + pr[2] + pr[3] + pr[4] + pr[6]) / st);
        else
            printf("probe  cycles per slab step and wavefront (%.0f steps sampled): wait for the slab %.0f, barrier %.0f, issue of the next slab %.0f, step word + fragments + matrix cores %.0f, between steps %.0f: %.0f in all\n",
                   st, pr[0] / st, pr[1] / st, pr[2] / st, pr[3] / st, pr[4] / st, (pr[0] + pr[1] + pr[2] + pr[3] + pr[4]) / st);
#endif
    }
    return 0;
}
