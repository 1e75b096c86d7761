// pg_hip_solve_ops.h -- the three solve-side operators of the platform table as device kernels (spmv, vecadd, sptrsv).
// Included by pg_hip_platform.hip.
#pragma once

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
