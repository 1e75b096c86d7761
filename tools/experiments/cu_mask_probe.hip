// Which CUs does a stream created with hipExtStreamCreateWithCUMask use?  (to reserve a few CUs for the latency-critical
// GETRF workgroups while bulk update launches fill the rest)
//   hipcc -O3 --offload-arch=gfx950 -o cu_mask_probe cu_mask_probe.hip && ./cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void probe(unsigned *out)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // busy for a while so that the grid spreads over every CU the queue may use
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 200000)
        ;
    if (threadIdx.x == 0)
        out[blockIdx.x] = ((xcc & 0xF) << 16) | (hw & 0xFFFF);
}
static void run(hipStream_t s, const char *what)
{
    const int n = 4096;
    unsigned *d;
    hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 65536, s, d);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(n);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> cus;
    int per_xcc[16] = {0};
    for (unsigned v : h)
    {
        // HW_ID: [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se
        unsigned key = (v >> 16) << 16 | ((v >> 8) & 0xFF);
        if (cus.insert(key).second)
            per_xcc[v >> 16]++;
    }
    printf("%-40s distinct CUs %3zu | per XCC:", what, cus.size());
    for (int i = 0; i < 8; i++)
        printf(" %d", per_xcc[i]);
    printf("\n");
    hipFree(d);
}
int main()
{
    hipStream_t s0;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    run(s0, "plain stream");
    for (int variant = 0; variant < 3; variant++)
    {
        uint32_t mask[8];
        for (int i = 0; i < 8; i++)
            mask[i] = 0xFFFFFFFFu;
        const char *what;
        if (variant == 0) { mask[7] &= 0x00FFFFFFu; what = "mask without bits 248..255"; }
        else if (variant == 1) { mask[0] &= 0xFFFFFF00u; what = "mask without bits 0..7"; }
        else { for (int i = 0; i < 8; i++) mask[i] &= 0x7FFFFFFFu; what = "mask without bit 31 of every word"; }
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
        if (e != hipSuccess) { printf("%s: %s\n", what, hipGetErrorString(e)); continue; }
        run(s, what);
        hipStreamDestroy(s);
    }
    return 0;
}
