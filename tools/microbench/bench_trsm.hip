// Stand-alone timing of trsm_dense_f64_kernel (pg_hip_trsm_dense.h)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef double val_t;
typedef unsigned int u32;
typedef unsigned short u16;
#define PANGULU_TOL 1e-16
struct TrsmTaskD { const u32 *vptr; const u16 *vidx; const u32 *vmap; val_t *bval; const u32 *tptr; const u16 *tidx; const val_t *tval; u32 is_tstrf; u32 pad_; };
__device__ inline u32 ptr0(const u32 *p, int i) { return i == 0 ? 0u : p[i]; }
__device__ inline unsigned long long wave_sum(unsigned long long v) { for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64); return v; }
typedef double v4f64 __attribute__((ext_vector_type(4)));
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
#define MIRROR_MAP_BYTES 64
__device__ inline const unsigned short *mirror_map(const double *mirror, int nb) { return reinterpret_cast<const unsigned short *>(mirror + (size_t)nb * nb); }
__device__ inline int mirror_column_of(const u32 *sp, int ncols, u32 p) { int lo = 0, hi = ncols; while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (sp[mid] <= p) lo = mid; else hi = mid; } return lo; }
__device__ inline const char __attribute__((address_space(1))) *dg_scalar_base(const char __attribute__((address_space(1))) *p)
{
    unsigned long long v = (unsigned long long)p;
    asm("" : "+s"(v));
    return (const char __attribute__((address_space(1))) *)v;
}
__device__ inline unsigned dg_lane_offset(unsigned v)
{
    asm("" : "+v"(v));
    return v;
}
#include "../../pangulu_amd/csrc/platform/pg_hip_trsm_dense.h"
#include "../../pangulu_amd/csrc/platform/pg_hip_trsm_ring.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main(int argc, char **argv)
{
    int nb = 256, ntask = argc > 1 ? atoi(argv[1]) : 1024, tstrf = argc > 2 ? atoi(argv[2]) : 1, nlu = argc > 3 ? atoi(argv[3]) : 64;
    size_t mb = (size_t)nb * nb + 8; // values + occupancy map (all tiles live)
    double *pool;
    int nmir = ntask + nlu;
    CK(hipMalloc(&pool, sizeof(double) * mb * nmir));
    std::vector<double> h(mb);
    for (size_t i = 0; i < mb; i++) h[i] = ((double)rand() / RAND_MAX - 0.5) * 0.01;
    for (int i = 0; i < nb; i++) h[(size_t)i * nb + i] = 1.0;
    for (int c = 0; c < 16; c++) reinterpret_cast<unsigned short *>(h.data() + (size_t)nb * nb)[c] = 0xFFFF;
    for (int i = 0; i < nmir; i++) CK(hipMemcpy(pool + (size_t)i * mb, h.data(), sizeof(double) * mb, hipMemcpyHostToDevice));
    std::vector<TrsmDenseTaskD> T(ntask);
    for (int t = 0; t < ntask; t++) { T[t].b = pool + (size_t)t * mb; T[t].lu = pool + (size_t)(ntask + t % nlu) * mb; T[t].is_tstrf = tstrf; T[t].lu_map = 0; T[t].progress = nullptr; }
    TrsmDenseTaskD *dT; CK(hipMalloc(&dT, sizeof(TrsmDenseTaskD) * ntask));
    CK(hipMemcpy(dT, T.data(), sizeof(TrsmDenseTaskD) * ntask, hipMemcpyHostToDevice));
    std::vector<u32> W((size_t)ntask * (nb / 64)); // work list: every (task, slab)
    for (size_t i = 0; i < W.size(); i++) W[i] = (u32)((i / (nb / 64)) << 2) | (u32)(i % (nb / 64));
    u32 *dW; CK(hipMalloc(&dW, sizeof(u32) * W.size())); CK(hipMemcpy(dW, W.data(), sizeof(u32) * W.size(), hipMemcpyHostToDevice));
    int direct = argc > 4 ? atoi(argv[4]) : 1; // 1: barrier-free kernel, 0: LDS-staged, 2: ring kernel (requests ahead through LDS)
    CK(hipFuncSetAttribute((const void *)trsm_dense_ring_f64_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tr_lds_bytes(256)));
    if (direct == 2)
    {
        // the ring kernel against the direct one on fresh copies of the same blocks
        std::vector<double> r1((size_t)mb * ntask), r2((size_t)mb * ntask);
        hipLaunchKernelGGL(trsm_dense_direct_f64_kernel<16>, dim3(ntask * (nb / 64)), dim3(256), 0, 0, dT, dW);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(r1.data(), pool, sizeof(double) * mb * ntask, hipMemcpyDeviceToHost));
        for (int i = 0; i < ntask; i++) CK(hipMemcpy(pool + (size_t)i * mb, h.data(), sizeof(double) * mb, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(trsm_dense_ring_f64_kernel<16>, dim3(ntask * (nb / 64)), dim3(256), tr_lds_bytes(256), 0, dT, dW);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(r2.data(), pool, sizeof(double) * mb * ntask, hipMemcpyDeviceToHost));
        double worst = 0, big = 0;
        for (size_t i = 0; i < r1.size(); i++) { worst = std::max(worst, std::fabs(r1[i] - r2[i])); big = std::max(big, std::fabs(r1[i])); }
        printf("check  ring against direct (tstrf %d, %d tasks): max difference %.3e, largest entry %.3e  %s\n", tstrf, ntask, worst, big, worst <= 1e-14 * big ? "ok" : "WRONG");
        for (int i = 0; i < ntask; i++) CK(hipMemcpy(pool + (size_t)i * mb, h.data(), sizeof(double) * mb, hipMemcpyHostToDevice));
    }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; rep++)
    {
        CK(hipEventRecord(a));
        if (direct == 2)
            hipLaunchKernelGGL(trsm_dense_ring_f64_kernel<16>, dim3(ntask * (nb / 64)), dim3(256), tr_lds_bytes(256), 0, dT, dW);
        else if (direct)
            hipLaunchKernelGGL(trsm_dense_direct_f64_kernel<16>, dim3(ntask * (nb / 64)), dim3(256), 0, 0, dT, dW);
        else
            hipLaunchKernelGGL(trsm_dense_f64_kernel<16>, dim3(ntask * (nb / 64)), dim3(256), 0, 0, dT, nullptr, dW);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("tasks %d tstrf %d distinct LU %d: %.3f ms  %.2f us/task  %.2f TFLOP/s (nb^3 per task)\n", ntask, tstrf, nlu, ms, 1e3 * ms / ntask, (double)nb * nb * nb * ntask / ms / 1e9);
    }
    return 0;
}
