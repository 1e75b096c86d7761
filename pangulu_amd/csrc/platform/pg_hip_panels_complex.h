// pg_hip_panels_complex.h -- dense-mode PANEL kernels of the complex value types (CR64; CR32, whose mirrors are double as
// well): GETRF of a diagonal block and TSTRF / GESSM against its L\U image, on the two-plane mirrors of pg_hip_dense.h (real
// plane at `dense`, imaginary plane mirror_plane_stride(nb) doubles behind; both nb x nb column-major).
// (included by pg_hip_platform.hip after pg_hip_dense.h, complex types only.)
//
// What they replace: the reference densifies and calls cuSOLVER getrf / cuBLAS trsm for every value type
// (...0201000.cu:547-641); until round 3 the complex types ran their panels on the pattern-driven kernels, where a 256 x 256
// front block costs milliseconds (poisson3d(48) CR64: 137 ms per factorisation against 15.6 for R64 with the same updates
// on the matrix cores).  These are plain blocked kernels on the vector units -- complex arithmetic in registers, panels of 16
// through LDS, lanes along rows so that every access to the column-major planes is a contiguous run -- not MFMA kernels: a
// complex 16 x 16 x 16 product is four real ones on planes, and the panel kernels of a complex front are a few per cent of
// its flops; what they had to stop being is a chain of dependent index lookups.
// Arithmetic: the CPU kernels' (...0100000.c:57-209) right-looking elimination without pivoting, pivot clamp
// |Re p| < 1e-16 -> 1e-16 when DIVIDING (the stored diagonal keeps its value), in a blocked order; entries outside the
// symbolic pattern stay exactly zero (the pattern is closed under elimination).  Parity: 1e-12 (CR64) / 1e-5 (CR32) of the oracle.
#pragma once

struct ZGetrfTaskD
{
    double *dense; // the block's mirror holding its current values; overwritten by L\U (unit lower, U with its diagonal)
};
struct ZTrsmTaskD
{
    double *b;        // mirror of the block being solved, overwritten by the solution
    const double *lu; // L\U image of the diagonal block
    u32 is_tstrf, slab; // slab: which 64 rows (TSTRF) / 64 columns (GESSM) of the block this workgroup solves
};

#define ZP_PANEL 16
#define ZG_THREADS 512
#define ZT_THREADS 256

__device__ __forceinline__ void z_submul(double &cr, double &ci, double ar, double ai, double br, double bi) // c -= a b
{
    cr = __builtin_fma(-ar, br, cr);
    cr = __builtin_fma(ai, bi, cr);
    ci = __builtin_fma(-ar, bi, ci);
    ci = __builtin_fma(-ai, br, ci);
}
// 1 / clamp(p): the reciprocal the divisions by a pivot multiply with
__device__ __forceinline__ void z_pivot_rcp(double pr, double pi, double &rr, double &ri)
{
    if ((pr < 0 ? -pr : pr) < PANGULU_TOL)
    {
        pr = PANGULU_TOL;
        pi = 0.0;
    }
    const double d = pr * pr + pi * pi;
    rr = pr / d;
    ri = -pi / d;
}
__device__ __forceinline__ void z_mul(double &xr, double &xi, double br, double bi) // x *= b
{
    const double tr = xr * br - xi * bi, ti = xr * bi + xi * br;
    xr = tr;
    xi = ti;
}

// The substitutions below keep a row's (column's) sixteen entries in registers and take their triangular operand from LDS or
// L2.  Fully unrolled they are 120 dependent multiply-adds with 240 operand loads the compiler hoists as far as it can (188
// spilled registers in the first version): the outer index runs as a real loop instead, the entry it needs is picked from the
// registers by compares, and the inner sixteen steps carry a predicate.
#define Z_PICK(dst_r, dst_i, xr_, xi_, k_)                              \
    {                                                                   \
        dst_r = 0.0;                                                    \
        dst_i = 0.0;                                                    \
        _Pragma("unroll") for (int q_ = 0; q_ < ZP_PANEL; q_++) if (q_ == (k_)) \
        {                                                               \
            dst_r = xr_[q_];                                            \
            dst_i = xi_[q_];                                            \
        }                                                               \
    }

// ---------------------------------------------------------------------------------------------------------------
// GETRF.  grid = diagonal blocks, one workgroup of 512 threads each (256 registers per thread: at 1024 the trailing update spilled); dynamic LDS = 4 * 16 * nb doubles (panel and strip
// images, both planes; 128 KiB at nb = 256).  Per panel of 16 columns: (1) the panel (rows from its diagonal down) goes to
// LDS; (2) its diagonal tile is factorised by one wavefront, the rows below by substitution, one thread each; (3) back to
// the image; (4) the strip right of the diagonal tile: one thread per column, forward substitution with the unit-lower
// tile, into the image and into LDS; (5) trailing block -= panel x strip: thread (row lane, column group), rows along
// lanes, sixteen complex multiply-adds per entry from the two LDS images.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ZG_THREADS) void zgetrf_planes_kernel(const ZGetrfTaskD *__restrict__ tasks, int nb)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double *Pr = reinterpret_cast<double *>(smem_raw); // Pr[c * nb + r]: column c (0..15) of the panel, row r (absolute)
    double *Pi = Pr + ZP_PANEL * nb;
    double *Sr = Pi + ZP_PANEL * nb; // Sr[k * nb + c]: row k (0..15) of the strip, column c (absolute)
    double *Si = Sr + ZP_PANEL * nb;
    double *Dr = tasks[blockIdx.x].dense, *Di = Dr + mirror_plane_stride(nb);
    __shared__ double s_rcp[2 * ZP_PANEL]; // reciprocals of the panel's (clamped) pivots
    __shared__ unsigned s_gmap[16];        // s_gmap[tc] bit tr: tile (tr, tc) of the block holds pattern entries (closed under elimination)
    const int tid = threadIdx.x;
    if (tid < 16)
        s_gmap[tid] = tid < nb / 16 ? (unsigned)mirror_map(Dr, nb)[tid] : 0u;
    __syncthreads();
    // (the mirror of a diagonal block is cleared as a whole before its entries are scattered: a dead tile holds zeros and stays
    //  zero, so the map only saves work here -- most diagonal blocks of the lower tree levels are a few tiles wide)
    for (int k0 = 0; k0 < nb; k0 += ZP_PANEL)
    {
        const int tk = k0 >> 4;
        unsigned row_tiles = 0; // tile row tk: bit tc = tile (tk, tc) live
        for (int tc = 0; tc < nb / 16; tc++)
            row_tiles |= ((s_gmap[tc] >> tk) & 1u) << tc;
        const unsigned col_tiles = s_gmap[tk]; // tile column tk: bit tr
        // (1) panel columns k0 .. k0+15, rows k0 .. nb-1
        for (int e = tid; e < ZP_PANEL * (nb - k0); e += ZG_THREADS)
        {
            const int c = e / (nb - k0), r = k0 + e % (nb - k0);
            Pr[c * nb + r] = Dr[(size_t)(k0 + c) * nb + r];
            Pi[c * nb + r] = Di[(size_t)(k0 + c) * nb + r];
        }
        __syncthreads();
        // (2) the panel.  (a) One wavefront factorises the 16 x 16 diagonal tile in LDS -- lane (row, quarter of the columns),
        //     sixteen steps ordered by the wavefront's own in-order LDS queue, no workgroup barrier -- and leaves the reciprocals of
        //     the (clamped) pivots; (b) the rows below the tile, one thread each: x = a U11^-1 by substitution along the row.
        //     (The first version ran the sixteen steps over the whole panel with two workgroup barriers each: 923 us per launch.)
        if (tid < 64)
        {
            const int row = tid & 15, cg = tid >> 4;
            for (int j = 0; j < ZP_PANEL; j++)
            {
                double rr, ri;
                z_pivot_rcp(Pr[j * nb + k0 + j], Pi[j * nb + k0 + j], rr, ri);
                if (tid == 0)
                {
                    s_rcp[2 * j] = rr;
                    s_rcp[2 * j + 1] = ri;
                }
                const bool below = row > j;
                double lr = 0.0, li = 0.0;
                if (below)
                {
                    lr = Pr[j * nb + k0 + row];
                    li = Pi[j * nb + k0 + row];
                    z_mul(lr, li, rr, ri);
                }
                wave_lds_fence();
                if (below)
                {
                    if (cg == (j & 3))
                    {
                        Pr[j * nb + k0 + row] = lr;
                        Pi[j * nb + k0 + row] = li;
                    }
                    for (int c = j + 1 + ((cg - (j + 1)) & 3); c < ZP_PANEL; c += 4)
                    {
                        double xr = Pr[c * nb + k0 + row], xi = Pi[c * nb + k0 + row];
                        z_submul(xr, xi, lr, li, Pr[c * nb + k0 + j], Pi[c * nb + k0 + j]);
                        Pr[c * nb + k0 + row] = xr;
                        Pi[c * nb + k0 + row] = xi;
                    }
                }
                wave_lds_fence();
            }
        }
        __syncthreads();
        for (int r = k0 + ZP_PANEL + tid; r < nb; r += ZG_THREADS)
        {
            if (!((col_tiles >> (r >> 4)) & 1u))
                continue; // (zeros: nothing to solve)
            double xr[ZP_PANEL], xi[ZP_PANEL];
#pragma unroll
            for (int c = 0; c < ZP_PANEL; c++)
            {
                xr[c] = Pr[c * nb + r];
                xi[c] = Pi[c * nb + r];
            }
#pragma unroll 1
            for (int c = 0; c < ZP_PANEL; c++)
            {
                double vr, vi;
                Z_PICK(vr, vi, xr, xi, c)
                z_mul(vr, vi, s_rcp[2 * c], s_rcp[2 * c + 1]);
#pragma unroll
                for (int q = 0; q < ZP_PANEL; q++)
                {
                    if (q == c)
                    {
                        xr[q] = vr;
                        xi[q] = vi;
                    }
                    else if (q > c)
                        z_submul(xr[q], xi[q], vr, vi, Pr[q * nb + k0 + c], Pi[q * nb + k0 + c]); // U(c, q), q > c
                }
            }
#pragma unroll
            for (int c = 0; c < ZP_PANEL; c++)
            {
                Pr[c * nb + r] = xr[c];
                Pi[c * nb + r] = xi[c];
            }
        }
        __syncthreads();
        // (3) the factorised panel back into the image
        for (int e = tid; e < ZP_PANEL * (nb - k0); e += ZG_THREADS)
        {
            const int c = e / (nb - k0), r = k0 + e % (nb - k0);
            Dr[(size_t)(k0 + c) * nb + r] = Pr[c * nb + r];
            Di[(size_t)(k0 + c) * nb + r] = Pi[c * nb + r];
        }
        const int m0 = k0 + ZP_PANEL; // first row / column of the trailing block
        if (m0 >= nb)
            break;
        // (4) strip: rows k0 .. k0+15 of the columns right of the tile, Y = L11^-1 T (unit lower)
        for (int c = m0 + tid; c < nb; c += ZG_THREADS)
        {
            if (!((row_tiles >> (c >> 4)) & 1u))
                continue; // (zeros; the trailing update skips this column)
            double sr[ZP_PANEL], si[ZP_PANEL];
#pragma unroll
            for (int k = 0; k < ZP_PANEL; k++)
            {
                sr[k] = Dr[(size_t)c * nb + k0 + k];
                si[k] = Di[(size_t)c * nb + k0 + k];
            }
#pragma unroll 1
            for (int k = 0; k < ZP_PANEL - 1; k++)
            {
                double vr, vi;
                Z_PICK(vr, vi, sr, si, k)
#pragma unroll
                for (int q = 1; q < ZP_PANEL; q++)
                    if (q > k)
                        z_submul(sr[q], si[q], Pr[k * nb + k0 + q], Pi[k * nb + k0 + q], vr, vi);
            }
#pragma unroll
            for (int k = 0; k < ZP_PANEL; k++)
            {
                Dr[(size_t)c * nb + k0 + k] = sr[k];
                Di[(size_t)c * nb + k0 + k] = si[k];
                Sr[k * nb + c] = sr[k];
                Si[k * nb + c] = si[k];
            }
        }
        __syncthreads();
        // (5) trailing block: A(r, c) -= sum_k L(r, k) U(k, c), r, c >= m0
        {
            const int rl = tid & 63, cg = tid >> 6; // rows m0 + rl + 64 i, columns m0 + cg + 16 j
            for (int r = m0 + rl; r < nb; r += 64)
            {
                if (!((col_tiles >> (r >> 4)) & 1u))
                    continue; // (L(r, panel) = 0)
                double lr[ZP_PANEL], li[ZP_PANEL];
#pragma unroll
                for (int k = 0; k < ZP_PANEL; k++)
                {
                    lr[k] = Pr[k * nb + r];
                    li[k] = Pi[k * nb + r];
                }
                // (four columns per pass: their loads are issued together -- the loop is bound by memory latency otherwise)
                for (int c = m0 + cg; c < nb; c += 4 * (ZG_THREADS / 64))
                {
                    double xr[4], xi[4];
                    bool on[4];
#pragma unroll
                    for (int u = 0; u < 4; u++)
                    {
                        const int cc = c + u * (ZG_THREADS / 64);
                        on[u] = cc < nb && ((row_tiles >> (cc >> 4)) & 1u); // (U(panel, cc) = 0 otherwise)
                        xr[u] = on[u] ? Dr[(size_t)cc * nb + r] : 0.0;
                        xi[u] = on[u] ? Di[(size_t)cc * nb + r] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++)
                    {
                        const int cc = c + u * (ZG_THREADS / 64);
                        if (!on[u])
                            continue;
#pragma unroll
                        for (int k = 0; k < ZP_PANEL; k++)
                            z_submul(xr[u], xi[u], lr[k], li[k], Sr[k * nb + cc], Si[k * nb + cc]);
                        Dr[(size_t)cc * nb + r] = xr[u];
                        Di[(size_t)cc * nb + r] = xi[u];
                    }
                }
            }
        }
        __syncthreads(); // (the trailing block is in memory before the next panel is read; the LDS images are free)
    }
}

// ---------------------------------------------------------------------------------------------------------------
// TSTRF (X U = B: rows of the block are independent) and GESSM (L X = B, L unit lower: columns are).
// grid = (task, 64-wide slab) pairs, 256 threads (64 row lanes x 4 column groups; the loops over the rest of the block are
// chains of load -> sixteen multiply-adds -> store per entry, bound by memory latency: four entries in flight per thread;
// 1024 threads per workgroup were 2.5x slower on poisson3d(48): most blocks of a level are small); panels in which the
// slab has no live tile are skipped; dynamic LDS = 2 * 16 * nb doubles (a panel of the factor, both planes).
//   TSTRF, per panel p of 16 columns: U(0 .. 16p+15, panel) is not needed -- the right-looking form updates the columns
//     right of the panel instead: thread (row lane, quarter) solves its row's 16 entries of the panel in registers against
//     U11 (LDS), then B(row, c) -= sum_k x_k U(16p + k, c) for its quarter of the later columns, U's row panel from LDS.
//   GESSM, per panel p of 16 rows: the 16 x 64 piece X(panel, slab) is solved by one thread per column (unit lower tile of L
//     from LDS) and left in LDS; then thread (row, quarter of the slab's columns) updates the rows below: B(r, c) -= sum_k
//     L(r, 16p + k) X(k, c), its sixteen L entries in registers.
// Lanes run along rows in both: every access to the column-major planes is a contiguous run.
// The mirror of an off-diagonal block is only defined on the 16 x 16 tiles its occupancy map marks (densify clears live tiles
// only): a dead tile reads as zero and is never written -- the solution has the block's pattern.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ZT_THREADS) void ztrsm_planes_kernel(const ZTrsmTaskD *__restrict__ tasks, int nb)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double *Fr = reinterpret_cast<double *>(smem_raw); // TSTRF: Fr[k * nb + c] = U(p0 + k, c);  GESSM: Fr[k * 64 + c] = X(p0 + k, slab column c)
    double *Fi = Fr + ZP_PANEL * nb;
    const ZTrsmTaskD T = tasks[blockIdx.x];
    double *Br = T.b, *Bi = Br + mirror_plane_stride(nb);
    const double *Lr = T.lu, *Li = Lr + mirror_plane_stride(nb);
    const int tid = threadIdx.x;
    const int s0 = (int)T.slab * 64;
    __shared__ unsigned s_map[16]; // s_map[tc] bit tr: tile (tr, tc) of the block holds pattern entries
    if (tid < 16)
        s_map[tid] = tid < nb / 16 ? (unsigned)mirror_map(Br, nb)[tid] : 0u;
    __syncthreads();
    if (T.is_tstrf)
    {
        const int r = s0 + (tid & 63), cq = tid >> 6; // row r of the block, later columns c = cq (mod 4)
        const unsigned slab_rows = 0xFu << (s0 >> 4); // the slab's four row tiles
        for (int p0 = 0; p0 < nb; p0 += ZP_PANEL)
        {
            if (!(s_map[p0 >> 4] & slab_rows))
                continue; // (no row of the slab has entries in this panel: x = 0, nothing to update)
            // rows p0 .. p0+15 of U, columns p0 .. nb-1
            for (int e = tid; e < ZP_PANEL * (nb - p0); e += ZT_THREADS)
            {
                const int c = p0 + e / ZP_PANEL, k = e % ZP_PANEL;
                Fr[k * nb + c] = Lr[(size_t)c * nb + p0 + k];
                Fi[k * nb + c] = Li[(size_t)c * nb + p0 + k];
            }
            __syncthreads();
            double xr[ZP_PANEL], xi[ZP_PANEL];
            const unsigned tr = (unsigned)r >> 4;
            const bool lv = (s_map[p0 >> 4] >> tr) & 1u; // this row's tile of the panel
#pragma unroll
            for (int k = 0; k < ZP_PANEL; k++)
            {
                xr[k] = lv ? Br[(size_t)(p0 + k) * nb + r] : 0.0;
                xi[k] = lv ? Bi[(size_t)(p0 + k) * nb + r] : 0.0;
            }
            __syncthreads(); // (the four threads of a row all start from the unsolved entries: one of them writes the solution below)
#pragma unroll 1
            for (int k = 0; k < ZP_PANEL; k++)
            {
                double rr, ri, vr, vi;
                z_pivot_rcp(Fr[k * nb + p0 + k], Fi[k * nb + p0 + k], rr, ri);
                Z_PICK(vr, vi, xr, xi, k)
                z_mul(vr, vi, rr, ri);
#pragma unroll
                for (int q = 0; q < ZP_PANEL; q++)
                {
                    if (q == k)
                    {
                        xr[q] = vr;
                        xi[q] = vi;
                    }
                    else if (q > k)
                        z_submul(xr[q], xi[q], vr, vi, Fr[k * nb + p0 + q], Fi[k * nb + p0 + q]);
                }
            }
            if (cq == 0 && lv)
            {
#pragma unroll
                for (int k = 0; k < ZP_PANEL; k++)
                {
                    Br[(size_t)(p0 + k) * nb + r] = xr[k];
                    Bi[(size_t)(p0 + k) * nb + r] = xi[k];
                }
            }
            // (four columns per pass: their loads are issued together)
            for (int c = p0 + ZP_PANEL + cq; lv && c < nb; c += 16)
            {
                double br[4], bi[4];
                bool on[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    const int cc = c + 4 * u;
                    on[u] = cc < nb && ((s_map[cc >> 4] >> tr) & 1u); // (outside the pattern the update is an exact zero)
                    br[u] = on[u] ? Br[(size_t)cc * nb + r] : 0.0;
                    bi[u] = on[u] ? Bi[(size_t)cc * nb + r] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    if (!on[u])
                        continue;
                    const int cc = c + 4 * u;
#pragma unroll
                    for (int k = 0; k < ZP_PANEL; k++)
                        z_submul(br[u], bi[u], xr[k], xi[k], Fr[k * nb + cc], Fi[k * nb + cc]);
                    Br[(size_t)cc * nb + r] = br[u];
                    Bi[(size_t)cc * nb + r] = bi[u];
                }
            }
            __syncthreads(); // (the factor panel is replaced next; a row's later columns were written by its own four threads only)
        }
    }
    else
    {
        unsigned slab_cols_rows = 0; // row tiles in which the slab's four column tiles have entries
        for (int t = 0; t < 4; t++)
            slab_cols_rows |= s_map[(s0 >> 4) + t];
        for (int p0 = 0; p0 < nb; p0 += ZP_PANEL)
        {
            if (!((slab_cols_rows >> (p0 >> 4)) & 1u))
                continue; // (the slab has no entries in this panel's rows: X(panel, slab) = 0)
            // X(panel rows, slab columns): one thread per column, forward substitution with the unit-lower tile L(p0.., p0..)
            if (tid < 64)
            {
                const int c = s0 + tid;
                const bool lv = (s_map[c >> 4] >> (p0 >> 4)) & 1u;
                double xr[ZP_PANEL], xi[ZP_PANEL];
#pragma unroll
                for (int k = 0; k < ZP_PANEL; k++)
                {
                    xr[k] = lv ? Br[(size_t)c * nb + p0 + k] : 0.0;
                    xi[k] = lv ? Bi[(size_t)c * nb + p0 + k] : 0.0;
                }
#pragma unroll 1
                for (int k = 0; k < ZP_PANEL - 1; k++)
                {
                    double vr, vi;
                    Z_PICK(vr, vi, xr, xi, k)
#pragma unroll
                    for (int q = 1; q < ZP_PANEL; q++)
                        if (q > k)
                            z_submul(xr[q], xi[q], Lr[(size_t)(p0 + k) * nb + p0 + q], Li[(size_t)(p0 + k) * nb + p0 + q], vr, vi);
                }
#pragma unroll
                for (int k = 0; k < ZP_PANEL; k++)
                {
                    if (lv)
                    {
                        Br[(size_t)c * nb + p0 + k] = xr[k];
                        Bi[(size_t)c * nb + p0 + k] = xi[k];
                    }
                    Fr[k * 64 + tid] = xr[k];
                    Fi[k * 64 + tid] = xi[k];
                }
            }
            __syncthreads();
            // rows below the panel: thread (row lane, quarter of the slab's columns); four columns' loads are issued together
            const int rl = tid & 63, cq = tid >> 6;
            for (int r = p0 + ZP_PANEL + rl; r < nb; r += 64)
            {
                double lr[ZP_PANEL], li[ZP_PANEL];
#pragma unroll
                for (int k = 0; k < ZP_PANEL; k++)
                {
                    lr[k] = Lr[(size_t)(p0 + k) * nb + r];
                    li[k] = Li[(size_t)(p0 + k) * nb + r];
                }
                for (int c4 = cq; c4 < 64; c4 += 16)
                {
                    double br[4], bi[4];
                    bool on[4];
#pragma unroll
                    for (int u = 0; u < 4; u++)
                    {
                        const int cc = c4 + 4 * u;
                        const unsigned m = s_map[(s0 + cc) >> 4];
                        on[u] = ((m >> (r >> 4)) & 1u) && ((m >> (p0 >> 4)) & 1u); // (tile outside the pattern, or the panel's piece of this column is zero)
                        const size_t at = (size_t)(s0 + cc) * nb + r;
                        br[u] = on[u] ? Br[at] : 0.0;
                        bi[u] = on[u] ? Bi[at] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++)
                    {
                        if (!on[u])
                            continue;
                        const int cc = c4 + 4 * u;
                        const size_t at = (size_t)(s0 + cc) * nb + r;
#pragma unroll
                        for (int k = 0; k < ZP_PANEL; k++)
                            z_submul(br[u], bi[u], lr[k], li[k], Fr[k * 64 + cc], Fi[k * 64 + cc]);
                        Br[at] = br[u];
                        Bi[at] = bi[u];
                    }
                }
            }
            __syncthreads(); // (the rows of the next panel are complete in memory; the LDS piece is free)
        }
    }
}
