// Stand-alone timing of ssssm_dense_f64_kernel (pg_hip_dense.h): hipcc --offload-arch=gfx950 -O3 -o bench_dense bench_dense.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef double val_t;
typedef unsigned int u32;
typedef unsigned short u16;
struct BlkView { const u32 *ptr; const u16 *idx; val_t *val; };
struct SsssmTaskD { BlkView a, b; };
struct SsssmGroupD { BlkView c; const u32 *ucp; const u16 *uri; const u32 *uvi; val_t *uval; val_t *cdense; u32 task_begin, task_end; u32 atomic; u32 slab_mask; u32 live_tiles; u32 pad_; }; // = pg_hip_platform.hip
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
#include "../../pangulu_amd/csrc/platform/pg_hip_dense.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main(int argc, char **argv)
{
    int nb = 256, ngroups = argc > 1 ? atoi(argv[1]) : 1024, tpg = argc > 2 ? atoi(argv[2]) : 8, nmir = 4096, atomic = argc > 3 ? atoi(argv[3]) : 0, livebits = argc > 4 ? atoi(argv[4]) : 16, ndst = argc > 5 ? atoi(argv[5]) : 1024, nops = argc > 6 ? atoi(argv[6]) : 1500;
    size_t mb = (size_t)nb * nb + 8 + 16 * (size_t)nb; // values + occupancy map + saved diagonal tiles (pg_hip_dense_host.h)
    double *pool;
    CK(hipMalloc(&pool, sizeof(double) * mb * nmir));
    std::vector<double> h(mb * 64);
    for (auto &x : h) x = (double)rand() / RAND_MAX - 0.5;
    for (int m = 0; m < 64; m++) // occupancy maps: the first `livebits` K-slabs live in every row/column slab
    {
        unsigned short *map = reinterpret_cast<unsigned short *>(h.data() + (size_t)m * mb + (size_t)nb * nb);
        for (int c = 0; c < 16; c++) map[c] = c < livebits ? (unsigned short)((1u << livebits) - 1u) : 0;
    }
    for (int i = 0; i < nmir; i += 64) CK(hipMemcpy(pool + (size_t)i * mb, h.data(), sizeof(double) * mb * 64, hipMemcpyHostToDevice));
    std::vector<SsssmGroupD> G(ngroups);
    std::vector<SsssmTaskD> T((size_t)ngroups * tpg);
    for (int g = 0; g < ngroups; g++)
    {
        memset(&G[g], 0, sizeof(SsssmGroupD));
        G[g].cdense = pool + (size_t)((g / ((ngroups + ndst - 1) / ndst)) % 1024) * mb; // consecutive groups (chunks of one queue) share a destination
        G[g].task_begin = g * tpg; G[g].task_end = (g + 1) * tpg; G[g].atomic = atomic;
        for (int t = 0; t < tpg; t++)
        {
            memset(&T[(size_t)g * tpg + t], 0, sizeof(SsssmTaskD));
            T[(size_t)g * tpg + t].a.val = pool + (size_t)(1024 + rand() % nops) * mb;
            T[(size_t)g * tpg + t].b.val = pool + (size_t)(2560 + rand() % nops) * mb;
        }
    }
    SsssmGroupD *dG; SsssmTaskD *dT;
    CK(hipMalloc(&dG, sizeof(SsssmGroupD) * G.size())); CK(hipMalloc(&dT, sizeof(SsssmTaskD) * T.size()));
    CK(hipMemcpy(dG, G.data(), sizeof(SsssmGroupD) * G.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dT, T.data(), sizeof(SsssmTaskD) * T.size(), hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    int tiles = nb / DG_TILE;
    std::vector<u32> W((size_t)ngroups * tiles * tiles); // work list: every (group, tile)
    for (size_t i = 0; i < W.size(); i++) W[i] = (u32)((i / (tiles * tiles)) << 2) | (u32)(i % (tiles * tiles));
    u32 *dW; CK(hipMalloc(&dW, sizeof(u32) * W.size())); CK(hipMemcpy(dW, W.data(), sizeof(u32) * W.size(), hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; rep++)
    {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(ssssm_dense_f64_kernel, dim3(ngroups * tiles * tiles), dim3(256), 0, 0, dG, dT, nb, nullptr, nullptr, dW);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        double flop = 2.0 * nb * nb * nb * (double)ngroups * tpg;
        printf("groups %d tasks/group %d atomic %d: %.3f ms  %.2f TFLOP/s\n", ngroups, tpg, atomic, ms, flop / ms / 1e9);
    }
    return 0;
}
