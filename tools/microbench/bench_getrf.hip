// Stand-alone check + timing of the dense GETRF kernels on diagonal blocks that live in mirrors: getrf_tiled_f64_kernel
// (pg_hip_getrf_tiled.h) against getrf_pipe_f64_kernel (pg_hip_getrf_pipe.h).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -o bench_getrf.bin bench_getrf.hip
//   ./bench_getrf.bin [nblocks = 1] [nb = 256]
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
#define PG_DENSE_PANELS 1
#define PANGULU_TOL 1e-16
struct BlkView { const u32 *ptr; const u16 *idx; val_t *val; };
struct SsssmTaskD { BlkView a, b; double sign; u32 count; u32 has_map; unsigned short amap[16], bmap_t[16]; };
struct SsssmWorkD { val_t *cdense; u32 task_begin, task_end; u32 atomic, slab_mask; u32 tile, pad_; };
struct GetrfTaskD { const u32 *lcp; const u16 *lri; val_t *lval; const u32 *urp; const u16 *uci; val_t *uval; val_t *dense; u32 preloaded, invert_tiles, defer_gather; };
__device__ inline u32 ptr0(const u32 *p, int i) { return i == 0 ? 0u : p[i]; }
__device__ inline unsigned long long wave_sum(unsigned long long v) { for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64); return v; }
__device__ inline void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double real_t;
__host__ __device__ inline val_t v_make(real_t r) { return r; }
__constant__ int c_xcd_swizzle = 1;
__device__ inline unsigned logical_block_id(unsigned per_unit) { return blockIdx.x; }
#include "../../pangulu_amd/csrc/platform/pg_hip_dense.h"
// (from pg_hip_getrf_blocked.h: the two helpers the tiled and the pipe kernel use)
__device__ inline int owner_of(const u32 *ptr, int n, u32 p)
{
    int lo = 0, hi = n;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ptr[mid] <= p) lo = mid; else hi = mid; }
    return lo;
}
#define GETRF_STAMP(slot) if (dbg && tid == 0 && blockIdx.x == 0) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); dbg[slot] += now_ - stamp_; stamp_ = now_; }
#include "../../pangulu_amd/csrc/platform/pg_hip_getrf_tiled.h"
#include "../../pangulu_amd/csrc/platform/pg_hip_getrf_pipe.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char **argv)
{
    const int nblk = argc > 1 ? atoi(argv[1]) : 1, nb = argc > 2 ? atoi(argv[2]) : 256;
    const size_t mb = mirror_plane_stride(nb);
    // a dense, diagonally dominant block; full occupancy map; trivial (dense) pattern arrays for the flop count
    std::vector<double> h(mb, 0.0);
    unsigned long long s = 88172645463325252ull;
    for (int c = 0; c < nb; c++)
        for (int r = 0; r < nb; r++)
        {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            h[(size_t)c * nb + r] = (double)(s >> 11) / 9007199254740992.0 - 0.5 + (r == c ? 0.6 * nb : 0.0);
        }
    unsigned short *map = reinterpret_cast<unsigned short *>(h.data() + (size_t)nb * nb);
    for (int c = 0; c < 16; c++) map[c] = c < nb / 16 ? (unsigned short)((1u << (nb / 16)) - 1u) : 0;
    double *dA, *dB;
    CK(hipMalloc(&dA, sizeof(double) * mb * nblk)); CK(hipMalloc(&dB, sizeof(double) * mb * nblk));
    std::vector<u32> cp(nb + 1);
    for (int c = 0; c <= nb; c++) cp[c] = 0; // (empty pattern arrays: the flop count is not what is timed)
    u32 *dcp; CK(hipMalloc(&dcp, sizeof(u32) * (nb + 1))); CK(hipMemcpy(dcp, cp.data(), sizeof(u32) * (nb + 1), hipMemcpyHostToDevice));
    unsigned long long *dflop; CK(hipMalloc(&dflop, 64 + 8 * 64)); CK(hipMemset(dflop, 0, 64 + 8 * 64));
    unsigned long long *ddbg = getenv("GP_STAMPS") ? dflop + 8 : nullptr;
    std::vector<GetrfTaskD> TA(nblk), TB(nblk);
    for (int b = 0; b < nblk; b++)
    {
        TA[b] = GetrfTaskD{dcp, nullptr, nullptr, dcp, nullptr, nullptr, dA + (size_t)b * mb, 1u, 1u, 1u};
        TB[b] = TA[b]; TB[b].dense = dB + (size_t)b * mb;
    }
    GetrfTaskD *dTA, *dTB;
    CK(hipMalloc(&dTA, sizeof(GetrfTaskD) * nblk)); CK(hipMalloc(&dTB, sizeof(GetrfTaskD) * nblk));
    CK(hipMemcpy(dTA, TA.data(), sizeof(GetrfTaskD) * nblk, hipMemcpyHostToDevice));
    CK(hipMemcpy(dTB, TB.data(), sizeof(GetrfTaskD) * nblk, hipMemcpyHostToDevice));
    const size_t lds_t = gt_lds_bytes(nb), lds_p = gp_lds_bytes(nb);
    CK(hipFuncSetAttribute((const void *)getrf_tiled_f64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t));
    CK(hipFuncSetAttribute((const void *)getrf_pipe_f64_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gp_lds_bytes(256)));
    CK(hipFuncSetAttribute((const void *)getrf_pipe_f64_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gp_lds_bytes(256)));
    auto reset = [&]() { for (int b = 0; b < nblk; b++) { CK(hipMemcpy(dA + (size_t)b * mb, h.data(), sizeof(double) * mb, hipMemcpyHostToDevice)); CK(hipMemcpy(dB + (size_t)b * mb, h.data(), sizeof(double) * mb, hipMemcpyHostToDevice)); } };
    auto run_tiled = [&]() { hipLaunchKernelGGL(getrf_tiled_f64_kernel, dim3(nblk), dim3(GT_THREADS), lds_t, 0, dTA, nb, dflop, (unsigned long long *)nullptr); };
    auto run_pipe = [&]() {
        if (nb == 256) hipLaunchKernelGGL(getrf_pipe_f64_kernel<16>, dim3(nblk), dim3(GP_THREADS), lds_p, 0, dTB, nb, dflop, ddbg);
        else hipLaunchKernelGGL(getrf_pipe_f64_kernel<8>, dim3(nblk), dim3(GP_THREADS), lds_p, 0, dTB, nb, dflop, ddbg);
    };
    // ---- check: both images (factors off the diagonal tiles, tile inverses on them) and the saved diagonal tiles
    reset();
    run_tiled(); run_pipe();
    CK(hipDeviceSynchronize());
    std::vector<double> a(mb), b(mb);
    CK(hipMemcpy(a.data(), dA, sizeof(double) * mb, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), dB, sizeof(double) * mb, hipMemcpyDeviceToHost));
    double worst = 0, big = 0, worst_saved = 0;
    for (size_t e = 0; e < (size_t)nb * nb; e++) { worst = fmax(worst, fabs(a[e] - b[e])); big = fmax(big, fabs(a[e])); }
    for (size_t e = (size_t)nb * nb + 8; e < (size_t)nb * nb + 8 + (size_t)16 * nb; e++) worst_saved = fmax(worst_saved, fabs(a[e] - b[e]));
    // ... and against a host LU without pivoting (off the diagonal tiles)
    std::vector<double> ref(h.begin(), h.begin() + (size_t)nb * nb);
    for (int k = 0; k < nb; k++)
    {
        const double p = ref[(size_t)k * nb + k];
        for (int r = k + 1; r < nb; r++) ref[(size_t)k * nb + r] /= p;
        for (int c = k + 1; c < nb; c++)
        {
            const double u = ref[(size_t)c * nb + k];
            for (int r = k + 1; r < nb; r++) ref[(size_t)c * nb + r] -= ref[(size_t)k * nb + r] * u;
        }
    }
    double worst_ref = 0;
    for (int c = 0; c < nb; c++)
        for (int r = 0; r < nb; r++)
            if ((r >> 4) != (c >> 4))
                worst_ref = fmax(worst_ref, fabs(b[(size_t)c * nb + r] - ref[(size_t)c * nb + r]));
    printf("check  nb = %d: max |tiled - pipe| = %.3e on the image (largest entry %.3e), %.3e on the saved diagonal tiles; max |pipe - host LU| off the diagonal tiles = %.3e  %s\n",
           nb, worst, big, worst_saved, worst_ref, (worst < 1e-11 * big && worst_saved < 1e-11 * big && worst_ref < 1e-11 * big) ? "ok" : "WRONG");
    // ---- timing
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int which = 0; which < 2; which++)
    {
        float best = 1e30f;
        for (int rep = 0; rep < 10; rep++)
        {
            reset();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            if (which == 0) run_tiled(); else run_pipe();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = fminf(best, ms);
        }
        printf("time   %-28s %d block(s) of %d: best %8.1f us\n", which == 0 ? "getrf_tiled_f64_kernel" : "getrf_pipe_f64_kernel", nblk, nb, 1e3 * best);
        if (which == 1 && ddbg)
        {
            unsigned long long st[64];
            CK(hipMemcpy(st, ddbg, sizeof(st), hipMemcpyDeviceToHost));
            const double runs = 11.0, f = 1.0 / runs / 100.0; // (s_memtime ticks at 100 MHz: 10 ns) -> us per factorisation
            printf("stamps (us per factorisation; block 0): prologue %.1f | wavefront 0: wait A %.1f, LU %.1f, inverses %.1f, wait B %.1f, wait C %.1f | trailing wavefront 1: "
                   "trailing %.1f, wait B %.1f, finish %.1f, wait C %.1f, next diagonal %.1f, wait A %.1f\n",
                   st[0] * f, st[16] * f, st[17] * f, st[18] * f, st[19] * f, st[20] * f, st[24] * f, st[25] * f, st[26] * f, st[27] * f, st[28] * f, st[29] * f);
        }
    }
    return 0;
}
