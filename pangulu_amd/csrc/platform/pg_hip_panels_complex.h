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

// ---------------------------------------------------------------------------------------------------------------
// Round 4 (end): TSTRF / GESSM of the complex types on the matrix cores.
//
// ztrsm_planes_kernel above is right-looking on the vector units: per panel every entry right of (below) it makes a round trip
// to memory -- poisson3d(80) CR64, nb = 128: 95 ms of 288 per factorisation (profiles/r04y_cr64_poisson80_kernel_stats.csv),
// 0.96 ms per launch.  Here the solve is the R64 one (pg_hip_trsm_dense.h, trsm_dense_direct_body) on two planes: one wavefront
// per 16-row (TSTRF) / 16-column (GESSM) strip keeps ALL its solution tiles in registers, left-looking,
//     X_p = (B_p - sum_{q<p} X_q U_qp) inv(U_pp)           X_p = inv(L_pp) (B_p - sum_{q<p} L_pq X_q)
// every complex tile product as four real v_mfma_f64_16x16x4_f64 products on the planes (re -= Ar Xr, re += Ai Xi, im -= Ar Xi,
// im -= Ai Xr: the NEG bit of the instruction does the signs), factor tiles straight from the L\U image (L2) into operand
// registers one stage ahead, no LDS, no barriers.  The inverses of the image's 16 x 16 diagonal tiles come from
// zdiag_tile_inverse_kernel, which runs behind zgetrf_planes_kernel and leaves them in the slack behind the planes (the
// 16 * nb doubles a blocked R64 GETRF saves its diagonal tiles in; unused by the complex types): tile p at p * 256, column-major,
// upper part inv(U_pp), strictly lower part inv(L_pp) without its unit diagonal -- the image itself stays L\U, the sparsify job
// behind the factorisation reads it.  Inverting instead of substituting costs a few ulps on these tiles (pivots clamped like
// the factorisation clamps them); parity: 1e-12 / 1e-5 of the oracle like the rest.
// PANGULU_HIP_ZTRSM_DIRECT=0 selects ztrsm_planes_kernel.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void zdiag_tile_inverse_kernel(const ZGetrfTaskD *__restrict__ tasks, int nb)
{
    const int np = nb / 16;
    const int p = (int)(blockIdx.x % (unsigned)np);
    double *Dr = tasks[blockIdx.x / (unsigned)np].dense, *Di = Dr + mirror_plane_stride(nb);
    __shared__ double Tr[16][17], Ti[16][17]; // T[row][col]
    const int lane = threadIdx.x;
    const size_t base = (size_t)(16 * p) * nb + 16 * p;
    for (int i = lane; i < 256; i += 64)
    {
        Tr[i & 15][i >> 4] = Dr[base + (size_t)(i >> 4) * nb + (i & 15)];
        Ti[i & 15][i >> 4] = Di[base + (size_t)(i >> 4) * nb + (i & 15)];
    }
    __syncthreads();
    if (lane >= 16)
        return;
    const int c = lane;
    double xur[16], xui[16], xlr[16], xli[16];
    // U x = e_c
#pragma unroll
    for (int r = 15; r >= 0; r--)
    {
        double sr = (r == c) ? 1.0 : 0.0, si = 0.0;
#pragma unroll
        for (int k = 15; k > r; k--)
            z_submul(sr, si, Tr[r][k], Ti[r][k], xur[k], xui[k]);
        double rr, ri;
        z_pivot_rcp(Tr[r][r], Ti[r][r], rr, ri);
        z_mul(sr, si, rr, ri);
        xur[r] = (r > c) ? 0.0 : sr;
        xui[r] = (r > c) ? 0.0 : si;
    }
    // L x = e_c  (unit lower)
#pragma unroll
    for (int r = 0; r < 16; r++)
    {
        double sr = (r == c) ? 1.0 : 0.0, si = 0.0;
#pragma unroll
        for (int k = 0; k < r; k++)
            z_submul(sr, si, Tr[r][k], Ti[r][k], xlr[k], xli[k]);
        xlr[r] = (r < c) ? 0.0 : sr;
        xli[r] = (r < c) ? 0.0 : si;
    }
    double *outr = Dr + (size_t)nb * nb + MIRROR_MAP_BYTES / sizeof(double) + (size_t)p * 256, *outi = outr + mirror_plane_stride(nb);
#pragma unroll
    for (int r = 0; r < 16; r++)
    {
        outr[c * 16 + r] = (r <= c) ? xur[r] : xlr[r];
        outi[c * 16 + r] = (r <= c) ? xui[r] : xli[r];
    }
}

// grid = (block, 64-wide slab) items, 256 threads = four wavefronts, each solving its own 16-wide strip; NP = nb / 16.
// Register layout as in trsm_dense_direct_body: TSTRF accumulates X^T tiles (register g of lane l = X(o0 + (l & 15), 16p + (l >> 4) + 4g)),
// GESSM X tiles (register g = X(16p + (l >> 4) + 4g, o0 + (l & 15))): a finished tile is the B operand of the later panels as it stands.
template <int NP>
__global__ __launch_bounds__(256) void ztrsm_direct_kernel(const ZTrsmTaskD *__restrict__ tasks)
{
    constexpr int nb = NP * 16;
    const ZTrsmTaskD T = tasks[blockIdx.x];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int o0 = (int)T.slab * 64 + wave * 16;
    double *__restrict__ Bm = T.b;
    const double *__restrict__ LU = T.lu;
    const bool tstrf = T.is_tstrf != 0;
    const size_t plane_bytes = mirror_plane_stride(nb) * sizeof(double);
    unsigned my_lv = 0; // bit p: tile p of this strip holds pattern entries
    {
        const unsigned short *map = mirror_map(Bm, nb);
        const int strip = o0 >> 4;
        if (tstrf)
        {
            for (int c = 0; c < NP; c++)
                my_lv |= (((unsigned)map[c] >> strip) & 1u) << c;
        }
        else
            my_lv = map[strip];
        my_lv = (unsigned)__builtin_amdgcn_readfirstlane((int)my_lv);
    }
    if (my_lv == 0)
        return;
    // which factor tiles hold pattern entries: the map densify left behind the image's real plane is the diagonal block's SYMBOLIC
    // pattern (fill included), so it describes L\U as well; a product with an empty factor tile is an exact no-op and is left out
    // (leaf-level diagonal blocks are sparse at tile granularity)
    unsigned fm[NP];
    {
        const unsigned short *fmap = mirror_map(LU, nb);
#pragma unroll
        for (int c = 0; c < NP; c++)
            fm[c] = (unsigned)__builtin_amdgcn_readfirstlane((int)fmap[c]);
    }
#define ZT_LIVE_Q(out_, p_)                                                         \
    {                                                                               \
        unsigned fl_ = 0;                                                           \
        if (tstrf)                                                                  \
            fl_ = fm[p_]; /* tile column p of the image: bit q = rows 16q.. */      \
        else                                                                        \
        {                                                                           \
            _Pragma("unroll") for (int q_ = 0; q_ < NP; q_++)                       \
                fl_ |= ((fm[q_] >> (p_)) & 1u) << q_; /* tile row p: bit q = columns 16q.. */ \
        }                                                                           \
        (out_) = my_lv & fl_;                                                       \
    }
    typedef const char __attribute__((address_space(1))) *gbytes;
    typedef const double __attribute__((address_space(1))) *gdouble_c;
    typedef double __attribute__((address_space(1))) *gdouble;
    // factor tile (q, p), q < p:  TSTRF  A'[i = c][k] = U(16q + k, 16p + c);  GESSM  A[i = r][k] = L(16p + r, 16q + k)
    const unsigned a_voff = (unsigned)(tstrf ? l15 * nb + l4 : l4 * nb + l15) * 8u;
#define ZT_A_LOAD(dr_, di_, q_, p_)                                                                                                  \
    {                                                                                                                                \
        const gbytes ab_ = (gbytes)LU + (tstrf ? (size_t)(16 * (p_)) * nb + 16 * (q_) : (size_t)(16 * (q_)) * nb + 16 * (p_)) * 8;   \
        _Pragma("unroll") for (int kq_ = 0; kq_ < 4; kq_++)                                                                          \
        {                                                                                                                            \
            const gbytes sb_ = ab_ + (tstrf ? (size_t)(4 * kq_) : (size_t)(4 * kq_) * nb) * 8;                                       \
            (dr_)[kq_] = *(gdouble_c)(dg_scalar_base(sb_) + dg_lane_offset(a_voff));                                                 \
            (di_)[kq_] = *(gdouble_c)(dg_scalar_base(sb_ + plane_bytes) + dg_lane_offset(a_voff));                                   \
        }                                                                                                                            \
    }
    // inverted diagonal tile p (zdiag_tile_inverse_kernel's layout: [c * 16 + r]):
    //   TSTRF  A''[i = c'][k = c] = invU(c, c') -> [i * 16 + k];   GESSM  A[i = r'][k = r] = invL(r', r) -> [k * 16 + i]
    const unsigned d_voff = (unsigned)(tstrf ? l15 * 16 + l4 : l4 * 16 + l15) * 8u;
#define ZT_D_LOAD(dr_, di_, p_)                                                                                                      \
    {                                                                                                                                \
        const gbytes db_ = (gbytes)LU + ((size_t)nb * nb + MIRROR_MAP_BYTES / sizeof(double) + (size_t)(p_) * 256) * 8;              \
        _Pragma("unroll") for (int kq_ = 0; kq_ < 4; kq_++)                                                                          \
        {                                                                                                                            \
            const gbytes sb_ = db_ + (tstrf ? (size_t)(4 * kq_) : (size_t)(4 * kq_) * 16) * 8;                                       \
            (dr_)[kq_] = *(gdouble_c)(dg_scalar_base(sb_) + dg_lane_offset(d_voff));                                                 \
            (di_)[kq_] = *(gdouble_c)(dg_scalar_base(sb_ + plane_bytes) + dg_lane_offset(d_voff));                                   \
        }                                                                                                                            \
    }
    const unsigned x_voff = (unsigned)(tstrf ? l4 * nb + l15 : l15 * nb + l4) * 8u;
#define ZT_X(plane_, p_, g_)                                                                                                         \
    (*(gdouble)(dg_scalar_base((gbytes)Bm + (tstrf ? (size_t)(16 * (p_) + 4 * (g_)) * nb + o0 : (size_t)o0 * nb + 16 * (p_) + 4 * (g_)) * 8 + \
                               ((plane_) ? plane_bytes : 0)) + dg_lane_offset(x_voff)))
    v4f64 xr[NP], xi[NP];
#pragma unroll
    for (int p = 0; p < NP; p++)
#pragma unroll
        for (int g = 0; g < 4; g++)
        {
            const bool lv = (my_lv >> p) & 1u;
            xr[p][g] = lv ? ZT_X(0, p, g) : 0.0;
            xi[p][g] = lv ? ZT_X(1, p, g) : 0.0;
        }
    // factor tiles in flight: two register sets used in turn (the loops unroll: the set index is static); the inverted diagonal
    // tile and the first factor tile of a panel are requested at the end of the panel before, ahead of its diagonal multiply
    double atr[2][4], ati[2][4], adr[2][4], adi[2][4];
    bool fetched = false;
    int set0 = 0;
#pragma unroll
    for (int p = 0; p < NP; p++)
    {
        const int st = set0;
        set0 = (set0 + p) & 1; // a panel has p stages
        if (!((my_lv >> p) & 1u))
        {
            fetched = false;
            continue; // (wavefront-uniform)
        }
        unsigned lq;
        ZT_LIVE_Q(lq, p)
        if (!fetched)
        {
            ZT_D_LOAD(adr[p & 1], adi[p & 1], p)
            if (p > 0 && (lq & 1u))
                ZT_A_LOAD(atr[st], ati[st], 0, p)
        }
        v4f64 pr[4], pi[4];
        pr[0] = xr[p];
        pi[0] = xi[p];
        pr[1] = pr[2] = pr[3] = (v4f64){0.0, 0.0, 0.0, 0.0};
        pi[1] = pi[2] = pi[3] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < p; q++)
        {
            const int cur = (st + q) & 1, nxt = cur ^ 1;
            __builtin_amdgcn_sched_barrier(0); // (fully unrolled, the scheduler otherwise hoists the loads of many stages: 1110 spilled registers at NP = 16)
            if (q + 1 < p && ((lq >> (q + 1)) & 1u))
                ZT_A_LOAD(atr[nxt], ati[nxt], q + 1, p)
            if ((lq >> q) & 1u)
            {
#pragma unroll
                for (int kq = 0; kq < 4; kq++)
                {
                    pr[kq] = __builtin_amdgcn_mfma_f64_16x16x4f64(atr[cur][kq], xr[q][kq], pr[kq], 0, 0, 1); // re -= Ar Xr
                    pi[kq] = __builtin_amdgcn_mfma_f64_16x16x4f64(atr[cur][kq], xi[q][kq], pi[kq], 0, 0, 1); // im -= Ar Xi
                    pr[kq] = __builtin_amdgcn_mfma_f64_16x16x4f64(ati[cur][kq], xi[q][kq], pr[kq], 0, 0, 0); // re += Ai Xi
                    pi[kq] = __builtin_amdgcn_mfma_f64_16x16x4f64(ati[cur][kq], xr[q][kq], pi[kq], 0, 0, 1); // im -= Ai Xr
                }
            }
        }
        fetched = false;
        __builtin_amdgcn_sched_barrier(0);
        if (p + 1 < NP && ((my_lv >> (p + 1)) & 1u))
        {
            unsigned lqn;
            ZT_LIVE_Q(lqn, p + 1)
            ZT_D_LOAD(adr[(p + 1) & 1], adi[(p + 1) & 1], p + 1)
            if (lqn & 1u)
                ZT_A_LOAD(atr[set0], ati[set0], 0, p + 1)
            fetched = true;
        }
        const v4f64 ar = (pr[0] + pr[1]) + (pr[2] + pr[3]), ai = (pi[0] + pi[1]) + (pi[2] + pi[3]);
        v4f64 yr = {0.0, 0.0, 0.0, 0.0}, yi = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kq = 0; kq < 4; kq++)
        {
            const int k = kq * 4 + l4;
            double dr = adr[p & 1][kq], di = adi[p & 1][kq];
            if (tstrf)
            {
                // invU(k, i) with i = l15: upper part of the tile
                dr = (k <= l15) ? dr : 0.0;
                di = (k <= l15) ? di : 0.0;
            }
            else
            {
                // invL(i, k) with i = l15: strictly lower part, unit diagonal
                dr = (l15 > k) ? dr : ((l15 == k) ? 1.0 : 0.0);
                di = (l15 > k) ? di : 0.0;
            }
            yr = __builtin_amdgcn_mfma_f64_16x16x4f64(dr, ar[kq], yr, 0, 0, 0); // re += Dr ar
            yi = __builtin_amdgcn_mfma_f64_16x16x4f64(dr, ai[kq], yi, 0, 0, 0); // im += Dr ai
            yr = __builtin_amdgcn_mfma_f64_16x16x4f64(di, ai[kq], yr, 0, 0, 1); // re -= Di ai
            yi = __builtin_amdgcn_mfma_f64_16x16x4f64(di, ar[kq], yi, 0, 0, 0); // im += Di ar
        }
        xr[p] = yr;
        xi[p] = yi;
    }
#pragma unroll
    for (int p = 0; p < NP; p++)
    {
        if (!((my_lv >> p) & 1u))
            continue;
#pragma unroll
        for (int g = 0; g < 4; g++)
        {
            ZT_X(0, p, g) = xr[p][g];
            ZT_X(1, p, g) = xi[p][g];
        }
    }
#undef ZT_X
#undef ZT_LIVE_Q
#undef ZT_D_LOAD
#undef ZT_A_LOAD
}
