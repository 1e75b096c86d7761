// Does a wavefront's v_mfma_f64_16x16x4_f64 stream leave the SIMD's vector ALU free for the OTHER wavefront of the SIMD?
// (gfx950: the f64 matrix rate equals the f64 vector rate, so the question is whether both share one datapath.)
//   hipcc -O3 --offload-arch=gfx950 -o mfma_f64_coissue mfma_f64_coissue.hip && ./mfma_f64_coissue
// One workgroup of 8 wavefronts on one CU = two per SIMD.  Wavefronts 0-3 run `role_a`, 4-7 `role_b`:
//   0 = exit at once, 1 = 2048 x 8 independent f64 MFMAs, 2 = 2048 x 64 dependent-free 32-bit integer VALU ops,
//   3 = 2048 x 64 v_fma_f64, 4 = 2048 x 16 ds_read_b64 + 16 VALU
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(int role_a, int role_b, unsigned long long *out, double *sink, int prio_b)
{
    __shared__ double lds[4096];
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? role_a : role_b;
    for (int i = threadIdx.x; i < 4096; i += 512)
        lds[i] = i;
    __syncthreads();
    if (wave >= 4 && prio_b)
        __builtin_amdgcn_s_setprio(3); // (does a higher wave priority get the second wavefront's VALU work in between the MFMAs?)
    unsigned long long t0 = __builtin_readcyclecounter();
    double r = 0;
    if (role == 1)
    {
        v4f64 acc[8];
        for (int i = 0; i < 8; i++)
            acc[i] = (v4f64){0, 0, 0, 0};
        double a = threadIdx.x, b = 1.0 / (1 + threadIdx.x);
        for (int it = 0; it < 2048; it++)
        {
#pragma unroll
            for (int i = 0; i < 8; i++)
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; i++)
            r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    else if (role == 2)
    {
        unsigned x[16];
        for (int i = 0; i < 16; i++)
            x[i] = threadIdx.x + i;
        for (int it = 0; it < 2048; it++)
        {
#pragma unroll
            for (int rep = 0; rep < 4; rep++)
#pragma unroll
                for (int i = 0; i < 16; i++)
                    x[i] = x[i] * 3u + (unsigned)it;
        }
        for (int i = 0; i < 16; i++)
            r += x[i];
    }
    else if (role == 3)
    {
        double x[16];
        for (int i = 0; i < 16; i++)
            x[i] = threadIdx.x + i;
        const double m = 1.0000001, c = 1e-9;
        for (int it = 0; it < 2048; it++)
        {
#pragma unroll
            for (int rep = 0; rep < 4; rep++)
#pragma unroll
                for (int i = 0; i < 16; i++)
                    x[i] = __builtin_fma(x[i], m, c);
        }
        for (int i = 0; i < 16; i++)
            r += x[i];
    }
    else if (role == 4)
    {
        unsigned idx = threadIdx.x & 63;
        for (int it = 0; it < 2048; it++)
        {
            double v[16];
#pragma unroll
            for (int i = 0; i < 16; i++)
                v[i] = lds[(idx + 64 * i + it) & 4095];
#pragma unroll
            for (int i = 0; i < 16; i++)
                r += v[i];
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0)
        out[wave] = role ? t1 - t0 : 0;
    if (r == 12345.678)
        sink[0] = r;
}

int main()
{
    unsigned long long *d, h[8];
    double *sink;
    hipMalloc(&d, sizeof(h));
    hipMalloc(&sink, 8);
    const char *names[] = {"idle", "mfma_f64 x16384", "int valu x131072", "fma_f64 x131072", "ds_read_b64 x32768 + add"};
    const int combos[][3] = {{1, 0, 0}, {2, 0, 0}, {3, 0, 0}, {4, 0, 0}, {1, 1, 0}, {1, 2, 0}, {1, 3, 0}, {1, 4, 0}, {2, 2, 0}, {3, 3, 0},
                             {1, 2, 1}, {1, 3, 1}, {1, 4, 1}, {1, 1, 1}};
    for (auto &c : combos)
    {
        for (int rep = 0; rep < 2; rep++)
        {
            hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, c[0], c[1], d, sink, c[2]);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("waves 0-3: %-28s waves 4-7: %-28s %s| clocks wave0 %9llu wave4 %9llu\n", names[c[0]], names[c[1]], c[2] ? "(s_setprio 3) " : "", h[0], h[4]);
    }
    return 0;
}
