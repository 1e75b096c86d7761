// What the f64 matrix cores of this chip sustain with nothing in their way: back-to-back v_mfma_f64_16x16x4_f64 on independent
// accumulators, operands in registers, random data, 1 / 2 / 4 wavefronts per SIMD on every CU, short and long (clock settles) runs.
// The "78.6 TFLOP/s" peak the roofline fractions are quoted against assumes 64 cycles per MFMA per SIMD at 2.4 GHz; this
// prints what is really there.  (Inline assembly: written with the builtin, the compiler keeps the accumulators in AGPRs
// inside the loop and copies all 64 registers to VGPRs and back every iteration -- 128 vector moves per 8 MFMAs.)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak.bin mfma_f64_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4f64 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define MFMA(acc_) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc_) : "v"(a), "v"(b))

__global__ __launch_bounds__(256) void mfma_loop(double *out, int iters, unsigned long long *clk)
{
    v4f64 c0, c1, c2, c3, c4, c5, c6, c7;
    const double a = 1.0 + 1e-9 * (threadIdx.x % 61), b = 0.999999 - 1e-9 * (threadIdx.x % 53);
    c0 = c1 = c2 = c3 = c4 = c5 = c6 = c7 = (v4f64){0.1, 0.2, 0.3, 0.4};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++)
    {
        MFMA(c0); MFMA(c1); MFMA(c2); MFMA(c3); MFMA(c4); MFMA(c5); MFMA(c6); MFMA(c7);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    v4f64 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0)
    {
        clk[2 * blockIdx.x] = t1 - t0;     // shader cycles
        clk[2 * blockIdx.x + 1] = r1 - r0; // 100 MHz ticks
    }
}

int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    double *out;
    unsigned long long *clk;
    CK(hipMalloc(&out, sizeof(double) * 256 * cus * 8));
    CK(hipMalloc(&clk, 16 * cus * 8));
    unsigned long long *hclk = (unsigned long long *)malloc(16 * cus * 8);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%s, %d CUs\n", p.name, cus);
    for (int wps = 1; wps <= 4; wps *= 2)
        for (int rep = 0; rep < 2; rep++)
        {
            const int iters = rep == 0 ? 20000 : 300000;
            const int grid = cus * wps;
            hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, 1000, clk); // warm
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, iters, clk);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(hclk, clk, 16 * grid, hipMemcpyDeviceToHost));
            double cyc = 0, ticks = 0, cmax = 0;
            for (int g = 0; g < grid; g++)
            {
                cyc += (double)hclk[2 * g];
                ticks += (double)hclk[2 * g + 1];
                if ((double)hclk[2 * g] > cmax)
                    cmax = (double)hclk[2 * g];
            }
            const double flop = (double)grid * 4 * iters * 8 * 2048.0;
            printf("%d workgroup(s) of 4 wavefronts per CU, %7d x 8 MFMAs per wavefront: %8.3f ms  %6.2f TFLOP/s  in-kernel clock %.3f GHz; one wavefront issues an MFMA every "
                   "%.1f cycles (mean over workgroups; slowest %.1f) -> %.1f cycles per MFMA per SIMD if %d share it\n",
                   wps, iters, ms, flop / ms / 1e9, cyc / ticks * 0.1, cyc / grid / ((double)iters * 8), cmax / ((double)iters * 8), cyc / grid / ((double)iters * 8) / wps, wps);
        }
    return 0;
}
