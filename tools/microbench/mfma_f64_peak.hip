// What the f64 matrix cores of this chip sustain with nothing in their way: back-to-back v_mfma_f64_16x16x4_f64 on independent
// accumulators, operands in registers, random data, 1 / 2 / 4 wavefronts per SIMD on every CU, short and long (clock settles) runs.
// The "78.6 TFLOP/s" peak the roofline fractions are quoted against assumes 2.4 GHz; this prints what is really there.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak.bin mfma_f64_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4f64 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double *out, int iters, unsigned long long *clk)
{
    v4f64 acc[NACC];
    const double a = 1.0 + 1e-9 * (threadIdx.x % 61), b = 0.999999 - 1e-9 * (threadIdx.x % 53);
#pragma unroll
    for (int j = 0; j < NACC; j++)
        acc[j] = (v4f64){0.1 * j, 0.2, 0.3, 0.4};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int j = 0; j < NACC; j++)
            acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int j = 0; j < NACC; j++)
        s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 7 && threadIdx.x == 0)
    {
        clk[0] = t1 - t0; // shader cycles
        clk[1] = r1 - r0; // 100 MHz ticks
    }
}

int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    double *out;
    unsigned long long *clk, hclk[2];
    CK(hipMalloc(&out, sizeof(double) * 256 * cus * 8));
    CK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%s, %d CUs\n", p.name, cus);
    for (int wps = 1; wps <= 4; wps *= 2)
        for (int rep = 0; rep < 2; rep++)
        {
            const int iters = rep == 0 ? 20000 : 400000; // ~5 ms and ~100+ ms per wavefront at one per SIMD
            const int grid = cus * wps;
            hipLaunchKernelGGL(mfma_loop<8>, dim3(grid), dim3(256), 0, 0, out, 1000, clk); // warm
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mfma_loop<8>, dim3(grid), dim3(256), 0, 0, out, iters, clk);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost));
            const double flop = (double)grid * 4 * iters * 8 * 2048.0;
            printf("%d wavefront(s) per SIMD, %7d x 8 MFMAs each: %8.3f ms  %6.2f TFLOP/s  in-kernel clock %.3f GHz, %.1f cycles per MFMA per SIMD\n", wps, iters, ms,
                   flop / ms / 1e9, (double)hclk[0] / (double)hclk[1] * 0.1, (double)hclk[0] / ((double)iters * 8 * wps));
        }
    return 0;
}
